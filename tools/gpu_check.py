#!/usr/bin/env python3
"""Developer check on a GPU box: HIP path vs the CPU oracle (f64 and long-double builds), per ray and
aggregated, for a few setups.  Prints a report; tests/ hold the asserted versions."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import solaraxionraytracing_amd as sa
from solaraxionraytracing_amd import _lib
from oracle.oracle import Oracle

FIELDS = ["pointdataX", "pointdataY", "pointdataR", "weights", "transmissionMagnet", "yawAngles", "reflect",
          "deviationDet", "pointdataXBefore", "pointdataYBefore", "energiesAx", "energiesPre", "transProbWindow",
          "transProbArgon", "pixvalsX", "shellNumber"]


def compare(name, full, n_rec=100_000, n_hist=1_000_000, flags=0, seed=3):
    print("=== %s  flags=%d" % (name, flags))
    with sa.RayTracer(full) as rt:
        t = time.time(); rec = rt.traceAxionWrapper(n_rec, seed=seed, flags=flags); t_rec = time.time() - t
        t = time.time(); img, summ = rt.trace_histogram(n_hist, seed=seed, flags=flags); t_h = time.time() - t
        t = time.time(); img, summ = rt.trace_histogram(n_hist, seed=seed, flags=flags); t_h2 = time.time() - t
    print("gpu records %.3fs, hist %.3fs / %.3fs (%.3g rays/s)" % (t_rec, t_h, t_h2, n_hist / t_h2))
    for variant in ("f64", "ld"):
        o = Oracle(full, variant)
        orec = o.trace_records(n_rec, seed=seed, flags=flags)
        for flag in ("passed", "passedTillWindow", "hitNickel"):
            ne = int((rec[flag] != orec[flag]).sum())
            print("  [%s] %-17s mismatches %d / %d (set: gpu %d oracle %d)" % (variant, flag, ne, n_rec, rec[flag].sum(), orec[flag].sum()))
        both = (rec["passed"] == 1) & (orec["passed"] == 1)
        for f in FIELDS:
            a, b = rec[f][both].astype(float), orec[f][both].astype(float)
            d = np.abs(a - b)
            rel = d / np.maximum(np.abs(b), 1e-300)
            print("  [%s] %-17s max abs %.3g  max rel %.3g" % (variant, f, d.max() if d.size else 0, rel.max() if rel.size else 0))
        # fields of rays that did not pass
        nb = ~both
        for f in ("energiesPre", "emratesPre", "deviationDet", "transmissionMagnet", "reflect"):
            d = np.abs(rec[f][nb] - orec[f][nb])
            print("  [%s] not-passed %-12s max abs %.3g" % (variant, f, d.max() if d.size else 0))
    o = Oracle(full)
    t = time.time(); oimg, osumm, used = o.trace_histogram(n_hist, seed=seed, flags=flags); t_o = time.time() - t
    print("oracle hist %.2fs on %d threads (%.3g rays/s)" % (t_o, used, n_hist / t_o))
    for k in summ:
        print("  %-22s gpu %.12g  oracle %.12g  rel %.3g" % (k, summ[k], osumm[k], (summ[k] - osumm[k]) / osumm[k] if osumm[k] else 0))
    print("  image: sum gpu %.12g oracle %.12g ; L1 diff / sum %.3g ; max pix rel %.3g" % (
        img.sum(), oimg.sum(), np.abs(img - oimg).sum() / oimg.sum(), np.abs(img - oimg).max() / oimg.max()))


if __name__ == "__main__":
    which = sys.argv[1:] or ["xmm", "llnl", "abrixas", "xmm_rot", "xmm_gas", "xmm_xray", "llnl_xray"]
    if "xmm" in which:
        compare("BabyIAXO/XMM/vacuum", sa.initFullSetup())
        compare("BabyIAXO/XMM/vacuum ignore-all", sa.initFullSetup(), flags=0b1111)
    if "llnl" in which:
        compare("CAST/LLNL/vacuum", sa.initFullSetup(_lib.ES_CAST, _lib.DK_INGRID2018, _lib.SK_VACUUM, _lib.TK_LLNL))
        compare("CAST/LLNL/gold", sa.initFullSetup(_lib.ES_CAST, _lib.DK_INGRID2018, _lib.SK_VACUUM, _lib.TK_LLNL, reflectivity="gold"))
    if "abrixas" in which:
        compare("CAST/Abrixas/vacuum", sa.initFullSetup(_lib.ES_CAST, _lib.DK_INGRID2017, _lib.SK_VACUUM, _lib.TK_ABRIXAS))
    if "xmm_rot" in which:
        full = sa.initFullSetup()
        full.setup.telescope_turned_y_deg = 0.05
        full.setup.telescope_turned_x_deg = 0.02
        full.setup.chip_x_max = full.setup.chip_y_max = 100.0
        compare("BabyIAXO/XMM rotated 0.02/0.05 deg chip 100", full, flags=0b1011)
    if "xmm_gas" in which:
        compare("BabyIAXO/XMM/gas", sa.initFullSetup(stage=_lib.SK_GAS))
    if "xmm_xray" in which:
        compare("BabyIAXO/XMM xrayTest", sa.initFullSetup(flags=_lib.CF_XRAY_TEST), flags=_lib.CF_XRAY_TEST)
    if "llnl_xray" in which:
        compare("CAST/LLNL xrayTest", sa.initFullSetup(_lib.ES_CAST, _lib.DK_INGRID2018, _lib.SK_VACUUM, _lib.TK_LLNL, flags=_lib.CF_XRAY_TEST), flags=_lib.CF_XRAY_TEST)
