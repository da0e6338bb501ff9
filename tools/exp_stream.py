#!/usr/bin/env python3
"""Acceptance test of a change of the random stream: the image and counters of a workload (default: BabyIAXO / XMM at 1e9 rays)
from the library with the NEW stream against the library with the OLD one, and - as the calibration of the statistic - the old
library against itself on another seed family.  Independent streams must agree within Monte-Carlo error:

  chi2 = sum over 8 x 8-pixel blocks with > 1e4 expected rays of (a - b)^2 / (var_a + var_b),  var = block sum x <w^2>/<w>
  (compound-Poisson variance of a sum of weights),  expected ndf +- 5 sqrt(2 ndf);  counters within 5 binomial sigma.

  python tools/exp_stream.py OLD.so [NEW.so|default] [rays] [babyiaxo_xmm|cast_llnl|babyiaxo_xmm_gas]
  (each library in its own subprocess; images under gpurun_out/)"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, json
sys.path.insert(0, %r)
import numpy as np
import solaraxionraytracing_amd as sa
from solaraxionraytracing_amd import _lib as L
tag, seed, n, out, wl = sys.argv[1], int(sys.argv[2]), int(float(sys.argv[3])), sys.argv[4], sys.argv[5]
full = {"babyiaxo_xmm": lambda: sa.initFullSetup(),
        "cast_llnl": lambda: sa.initFullSetup(L.ES_CAST, L.DK_INGRID2018, L.SK_VACUUM, L.TK_LLNL),
        "babyiaxo_xmm_gas": lambda: sa.initFullSetup(stage=L.SK_GAS)}[wl]()
with sa.RayTracer(full) as rt:
    img, s = rt.trace_histogram(n, seed=seed)
np.savez(out, img=img, keys=np.array(sorted(s)), vals=np.array([s[k] for k in sorted(s)]))
print(tag, "seed", seed, "passed", int(s["N_PASSED"]), "flux %%.9e" %% s["SUM_WEIGHTS"], flush=True)
''' % ROOT


WORKLOAD = "babyiaxo_xmm"


def run(lib, tag, seed, n):
    env = dict(os.environ)
    if lib != "default":
        env["SART_LIBSART"] = os.path.abspath(lib)
    out = os.path.join(ROOT, "gpurun_out", "stream_%s_%d.npz" % (tag, seed))
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.run([sys.executable, "-c", CHILD, tag, str(seed), str(n), out, WORKLOAD], env=env, check=True)
    d = np.load(out)
    return d["img"], dict(zip([str(k) for k in d["keys"]], d["vals"]))


def compare(name, a, sa_, b, sb, n):
    wa = sa_["SUM_WEIGHTS_SQ"] / sa_["SUM_WEIGHTS"]
    wb = sb["SUM_WEIGHTS_SQ"] / sb["SUM_WEIGHTS"]
    ba, bb = a.reshape(32, 8, 32, 8).sum(axis=(1, 3)), b.reshape(32, 8, 32, 8).sum(axis=(1, 3))
    mean_w = sa_["SUM_WEIGHTS"] / sa_["N_PASSED"]
    use = (ba + bb) / 2 > 1e4 * mean_w
    chi2 = float((((ba - bb) ** 2) / (ba * wa + bb * wb + 1e-300))[use].sum())
    ndf = int(use.sum())
    res = {"comparison": name, "chi2": chi2, "ndf": ndf, "pull_sigma": (chi2 - ndf) / np.sqrt(2.0 * ndf), "counters": {}}
    ok = abs(res["pull_sigma"]) < 5.0
    for k in ("N_REACHED_TELESCOPE", "N_SHELL_SELECTED", "N_HIT_NICKEL", "N_PASSED_TILL_WINDOW", "N_PASSED"):
        p = 0.5 * (sa_[k] + sb[k]) / n
        sig = np.sqrt(2.0 * n * p * (1.0 - p))
        res["counters"][k] = {"a": float(sa_[k]), "b": float(sb[k]), "pull_sigma": float((sa_[k] - sb[k]) / sig)}
        ok = ok and abs(res["counters"][k]["pull_sigma"]) < 5.0
    sig_flux = np.sqrt(sa_["SUM_WEIGHTS_SQ"] + sb["SUM_WEIGHTS_SQ"])
    res["flux"] = {"a": float(sa_["SUM_WEIGHTS"]), "b": float(sb["SUM_WEIGHTS"]), "pull_sigma": float((sa_["SUM_WEIGHTS"] - sb["SUM_WEIGHTS"]) / sig_flux)}
    res["within_5_sigma"] = bool(ok and abs(res["flux"]["pull_sigma"]) < 5.0)
    return res


def main():
    old = sys.argv[1]
    new = sys.argv[2] if len(sys.argv) > 2 else "default"
    n = int(float(sys.argv[3])) if len(sys.argv) > 3 else 1_000_000_000
    global WORKLOAD
    WORKLOAD = sys.argv[4] if len(sys.argv) > 4 else "babyiaxo_xmm"
    a, sa_ = run(old, "old", 299792458, n)
    c, sc = run(old, "old", 12345, n)
    b, sb = run(new, "new", 299792458, n)
    d, sd = run(new, "new", 12345, n)
    out = {"rays": n, "workload": WORKLOAD, "old": old, "new": new, "results": [
        compare("calibration: OLD stream, seed 299792458 vs seed 12345", a, sa_, c, sc, n),
        compare("OLD stream vs NEW stream, seed 299792458", a, sa_, b, sb, n),
        compare("OLD stream vs NEW stream, seed 12345", c, sc, d, sd, n),
        compare("NEW stream, seed 299792458 vs seed 12345", b, sb, d, sd, n)]}
    print(json.dumps(out, indent=1))
    for f in os.listdir(os.path.join(ROOT, "gpurun_out")):
        if f.startswith("stream_") and f.endswith(".npz"):
            os.remove(os.path.join(ROOT, "gpurun_out", f))
    raise SystemExit(0 if all(r["within_5_sigma"] for r in out["results"]) else 1)


if __name__ == "__main__":
    main()
