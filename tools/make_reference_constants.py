#!/usr/bin/env python3
"""Writes tests/golden/reference_constants.json: every numeric constant the reference's setup builders hold, read from the TEXT of
/root/reference/src/raytracer.nim (build container only: the reference does not travel).  Numbers only - keys are the reference's
field names, values are numbers / lists of numbers / enum ordinals; no source text is kept.

Why: the oracle and the HIP path are fed by the SAME product-built inputs (tests/conftest.py -> initFullSetup -> libsart_host.so), so
a wrong digit in a shell table is invisible to every GPU-vs-oracle test.  tests/test_reference_constants.py compares every field
of sart_host_new_full_setup with this file.

Sections read (line ranges as of the reference's commit): module constants :248-272, enums :18-47 / :59-64, toRad :322-332,
initMagnet :1098-1123, initPipes :1125-1155, initReflectivity :1158-1168 (kind + layers), initTelescope :1250-1348,
initTestXraySource :1350-1379, initDetectorInstallation :1381-1407, newDetectorSetup :1464-1490.
calcWindowVals (:1431-1462) is restated here and evaluated for the window parameters the reference uses.
Round 6: the sections, keys and default values of config/config_default.toml:1-50 (parsed, not copied) under "config_default_toml".

  python tools/make_reference_constants.py [--reference /root/reference] [--out tests/golden/reference_constants.json]
"""
import argparse
import json
import math
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NUM = r"[-+]?\d[\d_]*(?:\.\d+)?(?:[eE][-+]?\d+)?"


def strip_comments(text):
    return "\n".join(line.split("#", 1)[0] for line in text.splitlines())


def number(tok):
    """'225.0.mm' / '1485.mm' / '1e-12.GeV⁻¹' / '2+3+4' / '(sin(0.0.degToRad) * 7500.0).mm' -> float."""
    tok = tok.strip().rstrip(",").strip()
    m = re.fullmatch(r"\(\s*sin\(\s*(%s)\.degToRad\s*\)\s*\*\s*(%s)\s*\)(?:\.\w+)?" % (NUM, NUM), tok)
    if m:
        return math.sin(math.radians(float(m.group(1)))) * float(m.group(2))
    if re.fullmatch(r"\d+(?:\s*\+\s*\d+)+", tok):
        return float(sum(int(x) for x in tok.split("+")))
    m = re.match(r"(%s)(?:\.[A-Za-zμ°⁻¹²]+)?$" % NUM, tok)
    if not m:
        raise ValueError("not a number: %r" % tok)
    return float(m.group(1).replace("_", ""))


def proc_body(text, name):
    m = re.search(r"^proc %s\b.*?(?=^proc |^template |^defUnit|^import |\Z)" % re.escape(name), text, flags=re.S | re.M)
    if not m:
        raise KeyError(name)
    return m.group(0)


def branches(body):
    """{'esCAST': text, ...} for the `of a, b:` branches of the (first) case statement of a proc body."""
    out = {}
    parts = re.split(r"^\s*of\s+([\w, ]+):\s*$", body, flags=re.M)
    for names, txt in zip(parts[1::2], parts[2::2]):
        txt = re.split(r"^\s*else:\s*$", txt, flags=re.M)[0]
        for n in names.split(","):
            out[n.strip()] = txt
    return out


def fields(txt):
    """`key: value` pairs of an object constructor (top level: nested constructors and @[...] lists are values)."""
    out = {}
    for m in re.finditer(r"(\w+)\s*:\s*(@\[[^\]]*\](?:\.mapIt\([^)]*\))?|\w+\([^()]*\)|\([^()]*\([^()]*\)[^()]*\)(?:\.\w+)?|[^,\n()]+)", txt):
        out.setdefault(m.group(1), m.group(2).strip())
    return out


def seq(tok):
    inner = re.search(r"@\[(.*?)\]", tok, flags=re.S).group(1)
    return [number(x) for x in inner.replace("\n", " ").split(",") if x.strip()]


def enum_ordinals(text, name):
    m = re.search(r"^\s*%s\s*=\s*enum\s*\n(.*?)(?=^\s*\w+\*?\s*=\s*(?:enum|object|ref object))" % name, text, flags=re.S | re.M)
    return {n: i for i, n in enumerate(re.findall(r"^\s*(\w+)\b", m.group(1), flags=re.M))}


def calc_window_vals(radius, n_strips, open_ratio):
    """calcWindowVals (:1431-1462), restated."""
    total = math.pi * radius * radius
    area_strips = total * (1.0 - open_ratio)
    d_and_w = radius * 2.0 / (n_strips + 1.0)
    length_all = 0.0
    for i in range(int(round(n_strips / 2)) ):
        off = i * d_and_w + 0.5 * d_and_w
        length_all += math.sqrt(radius * radius - off * off) * 2.0
    length_all *= 2.0
    width = area_strips / length_all
    return width, d_and_w - width


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "reference_constants.json"))
    args = ap.parse_args()
    raw = open(os.path.join(args.reference, "src", "raytracer.nim"), encoding="utf-8").read()
    text = strip_comments(raw)
    out = {"what": "numeric constants of the reference's setup builders (src/raytracer.nim), made by tools/make_reference_constants.py"}

    const = {}
    for name in ("DistanceSunEarth", "RadiusSun", "NumberOfPointsSun", "RoomTemp", "mAxion", "g_aγ", "ChipXMin", "ChipXMax", "ChipYMin", "ChipYMax"):
        m = re.search(r"^\s*%s\s*=\s*(\S+)" % re.escape(name), text, flags=re.M)
        const[name.replace("γ", "gamma")] = number(m.group(1))
    out["constants"] = const
    enums = {n: enum_ordinals(text, n) for n in ("ExperimentSetupKind", "HoleType", "TelescopeKind", "StageKind", "WindowYearKind",
                                                   "ReflectivityKind", "DetectorSetupKind")}
    out["enums"] = enums

    out["windowYearDeg"] = {}
    for names, txt in branches(proc_body(text, "toRad")).items():
        out["windowYearDeg"][names] = number(re.search(r"degToRad\((%s)\)" % NUM, txt).group(1))

    out["magnet"] = {k: {f: number(v) for f, v in fields(re.search(r"Magnet\((.*)\)", t, flags=re.S).group(1)).items()}
                     for k, t in branches(proc_body(text, "initMagnet")).items()}

    pipes = {}
    for k, t in branches(proc_body(text, "initPipes")).items():
        if "Pipes(" not in t:
            continue
        body = re.search(r"Pipes\((.*)\)", t, flags=re.S).group(1)
        f = fields(body)
        row = {}
        for name in ("coldBoreToVT3", "vt3ToXRT"):
            inner = re.search(r"%s\s*:\s*Pipe\(([^()]*)\)" % name, body, flags=re.S).group(1)
            row[name] = {a: number(b) for a, b in fields(inner).items()}
        row["pipesTurned"] = number(f["pipesTurned"])
        row["distanceCBAxisXRTAxis"] = number(f["distanceCBAxisXRTAxis"])
        pipes[k] = row
    out["pipes"] = pipes

    tel = {}
    for k, t in branches(proc_body(text, "initTelescope")).items():
        if "Telescope(" not in t:
            continue
        body = re.search(r"Telescope\((.*)\)", t, flags=re.S).group(1)
        f = fields(body)
        row = {}
        for name in ("optics_entrance", "optics_exit", "allThickness", "allR1", "allXsep", "allAngles"):
            row[name] = seq(re.search(r"%s\s*:\s*(@\[.*?\])" % name, body, flags=re.S).group(1))
        for name in ("telescope_turned_x", "telescope_turned_y", "lMirror", "holeInOptics", "numberOfHoles"):
            row[name] = number(f[name])
        row["holeType"] = enums["HoleType"][f["holeType"].strip()]
        tel[k] = row
    out["telescope"] = tel

    refl = {}
    for k, t in branches(proc_body(text, "initReflectivity")).items():
        m = re.search(r"kind\s*:\s*(rk\w+)", t)
        if not m:
            continue
        row = {"kind": enums["ReflectivityKind"][m.group(1)]}
        lay = re.search(r"layers\s*:\s*(@\[[^\]]*\])", t)
        if lay:
            row["layers"] = seq(lay.group(1))
        refl[k] = row
    out["reflectivity"] = refl

    src = {}
    for k, t in branches(proc_body(text, "initTestXraySource")).items():
        f = fields(re.search(r"TestXraySource\((.*)\)", t, flags=re.S).group(1))
        row = {a: number(b) for a, b in f.items() if a not in ("active", "parallel")}
        row["parallel"] = 1.0 if f["parallel"].strip() == "true" else 0.0
        src[k] = row
    out["testSource"] = src

    inst = {}
    for k, t in branches(proc_body(text, "initDetectorInstallation")).items():
        if "DetectorInstallation(" not in t:
            continue
        inst[k] = {a: number(b) for a, b in fields(re.search(r"DetectorInstallation\((.*)\)", t, flags=re.S).group(1)).items()}
    out["detectorInstall"] = inst

    det = {}
    for k, t in branches(proc_body(text, "newDetectorSetup")).items():
        row = {}
        t = t.split("result.detectorWindowAperture")[0]   # (what follows the case statement belongs to every kind)
        for a, b in re.findall(r"result\.(\w+)\s*=\s*(\S+)", t):
            row[a] = float(enums["WindowYearKind"][b]) if a == "windowYear" else number(b)
        det[k] = row
    out["detector"] = det
    out["calcWindowVals"] = [{"radiusWindow": r, "numberOfStrips": n, "openApertureRatio": o, "width_dist": list(calc_window_vals(r, n, o))}
                             for r, n, o in sorted({(d["radiusWindow"], d["numberOfStrips"], d["openApertureRatio"]) for d in det.values()})]

    # config/config_default.toml:1-50 (BASELINE configs[0]'s plumbing): every section, key and default VALUE (numbers, booleans, file
    # and enum names) as data - comments and layout are not kept.  tests/test_config_io.py holds config.py to exactly this key set.
    import tomli
    with open(os.path.join(args.reference, "config", "config_default.toml"), "rb") as f:
        out["config_default_toml"] = tomli.load(f)

    n_numbers = sum(1 for _ in re.finditer(NUM, json.dumps(out)))
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote %s: %d numbers; shells %s" % (args.out, n_numbers, {k: len(v["allR1"]) for k, v in tel.items()}))


if __name__ == "__main__":
    main()
