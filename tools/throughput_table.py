#!/usr/bin/env python3
"""Throughput of the histogram path for a set of configurations (GPU box): 20 launches of 1e8 rays behind 5 untimed ones.
Prints a markdown table."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import solaraxionraytracing_amd as sa
from solaraxionraytracing_amd import _lib as L

def run(name, full, n=100_000_000, flags=None, reps=20):
    with sa.RayTracer(full) as rt:
        for k in range(5):   # the first launches after an idle gap run while the clock is still ramping up
            rt.trace_histogram(n, seed=2, ray_id_offset=k * n, flags=flags)
        rt.enable_kernel_timing(True)
        for k in range(reps):
            img, s = rt.trace_histogram(n, seed=1, ray_id_offset=k * n, flags=flags, accumulate=(k > 0))
        ms, nl = rt.kernel_timing()
    print("| %s | %.3f | %.3g | %.3f | %.3f |" % (name, ms / nl, n / (ms / nl) * 1e3, s["N_SHELL_SELECTED"] / s["N_RAYS"], s["N_PASSED"] / s["N_RAYS"]), flush=True)

print("| configuration | ms / 1e8 rays | rays/s | to mirrors | passed |\n|---|---|---|---|---|")
run("BabyIAXO / XMM / vacuum (configs[2])", sa.initFullSetup())
run("CAST / LLNL / gold (configs[1])", sa.initFullSetup(L.ES_CAST, L.DK_INGRID2018, L.SK_VACUUM, L.TK_LLNL, reflectivity="gold"))
run("CAST / LLNL / 4 coatings", sa.initFullSetup(L.ES_CAST, L.DK_INGRID2018, L.SK_VACUUM, L.TK_LLNL))
run("CAST / Abrixas", sa.initFullSetup(L.ES_CAST, L.DK_INGRID2017, L.SK_VACUUM, L.TK_ABRIXAS))
run("BabyIAXO / XMM / gas (configs[4])", sa.initFullSetup(stage=L.SK_GAS))
full = sa.initFullSetup(); full.setup.telescope_turned_y_deg = 0.1; full.setup.chip_x_max = full.setup.chip_y_max = 100.0
run("BabyIAXO / XMM rotated 0.1 deg, chip 100 mm, effective-area flags (configs[3])", full, flags=0b1011)
run("BabyIAXO / XMM X-ray test source", sa.initFullSetup(flags=L.CF_XRAY_TEST), flags=L.CF_XRAY_TEST)
