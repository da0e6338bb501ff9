"""The reference's configuration surface for the hot path: `config/config.toml` (keys of
config/config_default.toml:1-50; parsers raytracer.nim:984-1096) and the cligen flags of `main`
(raytracer.nim:2817-2850) -> the inputs of `initFullSetup`.  Host-side plumbing only.

Resolution of the three input files follows the reference: `[Resources].resourcePath` / `solarModelFile`,
`goldReflFile`, `llnlReflFile` (:2645-2647, :1170-1171, :1193-1194).  When a file is not there (the reference does not
ship them) the documented synthetic stand-in of tables.py is used and noted in ``FullRaytraceSetup.meta``.
"""
from __future__ import annotations

import os

from . import _lib, tables
from .raytracer import FullRaytraceSetup, initFullSetup

_ENUMS = {
    "experimentSetup": {"CAST": _lib.ES_CAST, "BabyIAXO": _lib.ES_BABYIAXO},                       # raytracer.nim:16-18
    "detectorSetup": {"InGrid2017": _lib.DK_INGRID2017, "InGrid2018": _lib.DK_INGRID2018,
                      "InGridIAXO": _lib.DK_INGRIDIAXO},                                              # :164-167
    "stageSetup": {"vacuum": _lib.SK_VACUUM, "gas": _lib.SK_GAS},                                    # :39-41
    "telescopeSetup": {"LLNL": _lib.TK_LLNL, "XMM": _lib.TK_XMM, "CustomBabyIAXO": _lib.TK_CUSTOM_BABYIAXO,
                       "Abrixas": _lib.TK_ABRIXAS, "Other": _lib.TK_OTHER},                          # :31-37
}


# Every section and key of config/config_default.toml:1-50 and who reads it here (tests/test_config_io.py holds this table to the
# key set of the reference's file, tests/golden/reference_constants.json["config_default_toml"]: a key the reference adds or drops
# turns that test red, a key nobody reads cannot hide in a sample file).
CONFIG_KEYS = {
    "Resources": {
        "resourcePath": "init_full_setup_from_config, read_opacity_file: directory of the input files (relative to the config file)",
        "outputPath": "init_full_setup_from_config -> FullRaytraceSetup.outpath; read_opacity_file: where the solar model CSV goes",
        "llnlEfficiency": "resolve_resources -> tables.llnl_effective_area(path): the DTU-thesis curve the LLNL plots overlay",
        "goldFilePrefix": "resolve_resources -> tools/convert_reflectivities_to_h5.py: the Henke download directory",
        "rawSolarModel": "read_opacity_file: the AGSS09 table",
        "solarModelFile": "init_full_setup_from_config: emission table CSV (raytracer.nim:2645-2647)",
        "llnlReflFile": "init_full_setup_from_config: 4-coating reflectivity H5 for tkLLNL (:1170-1171)",
        "goldReflFile": "init_full_setup_from_config: gold reflectivity H5 for the other telescopes (:1193-1194)",
    },
    "ReadOpacityFile": {
        "solarModelFile": "read_opacity_file: name of the CSV it writes",
        "opcdPath": "resolve_opcd_path: OPCD 3.3 directory (readOpacityFile.nim:114-117)",
    },
    "Setup": {k: "parse_setup (raytracer.nim:1024-1030)" for k in ("experimentSetup", "detectorSetup", "stageSetup", "telescopeSetup")},
    "Magnet": {k: "maybeParseMagnetConfig (:1032-1051)" for k in ("useConfig", "B", "radiusCB", "lengthColdbore", "lengthB", "pGasRoom", "tGas")},
    "TestXraySource": {k: "maybeParseTestXraySource (:1053-1076)" for k in (
        "useConfig", "active", "parallel", "energy", "distance", "radius", "offAxisUp", "offAxisLeft", "activity", "lengthCol")},
    "DetectorInstallation": {k: "maybeParseDetectorInstallation (:1078-1096)" for k in (
        "useConfig", "distanceDetectorXRT", "distanceWindowFocalPlane", "lateralShift", "transversalShift")},
}


def check_keys(cfg: dict) -> list:
    """Keys of a parsed config file that nothing here reads (typos, keys of a newer reference): returned, not raised - the reference's
    parsers ignore unknown keys too (parsetoml lookups by name)."""
    return ["[%s].%s" % (sec, k) for sec, body in cfg.items() if isinstance(body, dict)
            for k in body if k not in CONFIG_KEYS.get(sec, {})] + ["[%s]" % sec for sec in cfg if sec not in CONFIG_KEYS]


def resolve_resources(cfg: dict, base: str) -> dict:
    """Every `[Resources]` entry as the path the reference would open (resourcePath / name, :1012-1022; the two directories relative
    to the config file's directory), whether or not the file is there."""
    res = cfg.get("Resources", {})
    rdir = os.path.normpath(os.path.join(base, res.get("resourcePath", "../resources")))
    out = {"resourcePath": rdir, "outputPath": os.path.normpath(os.path.join(base, res.get("outputPath", "../out")))}
    for k in ("llnlEfficiency", "goldFilePrefix", "rawSolarModel", "solarModelFile", "llnlReflFile", "goldReflFile"):
        if res.get(k):
            out[k] = os.path.join(rdir, res[k])
    return out


def flags_from_cli(ignoreDetWindow=False, ignoreGasAbs=False, ignoreConvProb=False, ignoreReflection=False, xrayTest=False,
                   detectorInstall=False, magnet=False) -> int:
    """set[ConfigFlags] from the switches of `main` (raytracer.nim:2842-2849)."""
    f = 0
    if ignoreDetWindow: f |= _lib.CF_IGNORE_DET_WINDOW
    if ignoreGasAbs: f |= _lib.CF_IGNORE_GAS_ABS
    if ignoreConvProb: f |= _lib.CF_IGNORE_CONV_PROB
    if ignoreReflection: f |= _lib.CF_IGNORE_REFLECTION
    if xrayTest: f |= _lib.CF_XRAY_TEST
    if magnet: f |= _lib.CF_READ_MAGNET_CONFIG
    if detectorInstall: f |= _lib.CF_READ_DET_INSTALL_CONFIG
    return f


def parse_setup(cfg: dict):
    """parseSetup (raytracer.nim:1024-1030); an unknown name raises ValueError like parseEnum."""
    out = []
    for key in ("experimentSetup", "detectorSetup", "stageSetup", "telescopeSetup"):
        name = cfg["Setup"][key]
        if name not in _ENUMS[key]:
            raise ValueError("invalid enum value: %s = %r" % (key, name))
        out.append(_ENUMS[key][name])
    return tuple(out)


def resolve_opcd_path(cfg: dict, base: str):
    """`[ReadOpacityFile].opcdPath` (parseOpcdPath, readOpacityFile.nim:114-117).  The reference uses the string as it is
    (relative to the working directory); a relative path that is not there is also tried beside the config file."""
    p = (cfg.get("ReadOpacityFile", {}) or {}).get("opcdPath")
    if not p:
        return None
    if os.path.isabs(p) or os.path.isdir(p):
        return os.path.normpath(p)
    return os.path.normpath(os.path.join(base, p))


def load_config(path: str) -> dict:
    import tomli
    with open(path, "rb") as f:
        return tomli.load(f)


def init_full_setup_from_config(config_path: str, flags: int = 0, **overrides) -> FullRaytraceSetup:
    """initFullSetup(parseSetup()..., flags) driven by a config.toml (raytracer.nim:2852-2859)."""
    cfg = load_config(config_path)
    es, dk, sk, tk = parse_setup(cfg)
    base = os.path.dirname(os.path.abspath(config_path))
    res = cfg.get("Resources", {})
    paths = resolve_resources(cfg, base)
    rdir = paths["resourcePath"]

    magnet_cfg = source_cfg = install_cfg = None
    m = cfg.get("Magnet", {})
    if (flags & _lib.CF_READ_MAGNET_CONFIG) or m.get("useConfig", False):          # maybeParseMagnetConfig :1032-1051
        magnet_cfg = _lib.MagnetConfig(m["B"], m["radiusCB"], m["lengthColdbore"], m["lengthB"], m["pGasRoom"], m["tGas"])
    t = cfg.get("TestXraySource", {})
    if (flags & _lib.CF_XRAY_TEST) or t.get("useConfig", False):                   # maybeParseTestXraySource :1053-1076
        source_cfg = _lib.TestSourceConfig(int(t["active"]), int(t["parallel"]), t["energy"], t["distance"], t["radius"],
                                           t["offAxisUp"], t["offAxisLeft"], t["activity"], t["lengthCol"])
    d = cfg.get("DetectorInstallation", {})
    if (flags & _lib.CF_READ_DET_INSTALL_CONFIG) or d.get("useConfig", False):     # maybeParseDetectorInstallation :1078-1096
        install_cfg = _lib.DetectorInstallConfig(d["distanceDetectorXRT"], d["distanceWindowFocalPlane"], d["lateralShift"],
                                                 d["transversalShift"])

    kw = dict(magnet_cfg=magnet_cfg, source_cfg=source_cfg, install_cfg=install_cfg)
    notes = []
    solar = os.path.join(rdir, res.get("solarModelFile", "solar_model_dataframe.csv"))
    if os.path.exists(solar):
        kw["solar_model_csv"] = solar
    else:
        # the reference would run readOpacityFile first (README.org:22-55): with the OPCD files at hand the same table is made
        # on the GPU (emission kernel with the OPCD absorption coefficients) instead of being read back from the CSV
        opcd = resolve_opcd_path(cfg, base)
        if opcd and os.path.exists(os.path.join(opcd, "OPCD_3.3", "mono", "fm01.mesh")):
            kw["emission"] = "agss09-device"
            kw["opcd_path"] = opcd
            kw["opcd_optional"] = True   # an incomplete OPCD directory degrades to the table without the OPCD term (noted in meta)
            notes.append("solarModelFile %s not found: emission table from the AGSS09 model and the OPCD files in %s" % (solar, opcd))
        else:
            notes.append("solarModelFile %s not found: synthetic E1 emission table" % solar)
    refl_key = "llnlReflFile" if tk == _lib.TK_LLNL else "goldReflFile"
    refl = os.path.join(rdir, res.get(refl_key, ""))
    if res.get(refl_key) and os.path.exists(refl):
        kw["reflectivity"] = tables.read_reflectivity_h5(refl)
    else:
        notes.append("%s %s not found: synthetic Henke-derived reflectivity" % (refl_key, refl))
    kw.update(overrides)
    full = initFullSetup(es, dk, sk, tk, flags, **kw)
    full.outpath = paths["outputPath"]
    full.meta["config"] = config_path
    unknown = check_keys(cfg)
    if unknown:
        notes.append("config keys nothing reads: " + ", ".join(unknown))
    full.meta["notes"] = notes + list(full.meta.get("notes", []))
    return full
