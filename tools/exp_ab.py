#!/usr/bin/env python3
"""A/B timing of kernel builds: python tools/exp_ab.py lib1.so lib2.so ...  (each in its own subprocess, alternating).
Environment: SART_AB_RAYS (rays per launch, default 1e8), SART_AB_REPS (timed launches, default 5), SART_AB_ROUNDS (times the
list of libraries is gone through, default 1), SART_AB_WORKLOADS (comma list of BabyIAXO,CAST,gas,rot; default the first three)."""
import sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os
sys.path.insert(0, %r)
import solaraxionraytracing_amd as sa
from solaraxionraytracing_amd import _lib as L
N = int(float(os.environ.get("SART_AB_RAYS", "1e8")))
REPS = int(os.environ.get("SART_AB_REPS", "5"))
WL = os.environ.get("SART_AB_WORKLOADS", "BabyIAXO,CAST,gas").split(",")
def run(name, full, flags=None):
    if name not in WL:
        return
    with sa.RayTracer(full) as rt:
        for k in range(3):
            rt.trace_histogram(N, seed=2, ray_id_offset=k * N, flags=flags)     # clocks up
        rt.enable_kernel_timing(True)
        for k in range(REPS):
            img, s = rt.trace_histogram(N, seed=1, ray_id_offset=k * N, accumulate=(k > 0), flags=flags)
        ms, nl = rt.kernel_timing()
    print("  %%-10s %%.4f ms / %%.0e rays  flux %%.9e passed %%d" %% (name, ms / nl, N, s["SUM_WEIGHTS"], s["N_PASSED"]), flush=True)
run("BabyIAXO", sa.initFullSetup())
run("CAST", sa.initFullSetup(L.ES_CAST, L.DK_INGRID2018, L.SK_VACUUM, L.TK_LLNL, reflectivity="gold"))
run("gas", sa.initFullSetup(stage=L.SK_GAS))
if "rot" in WL:
    full = sa.initFullSetup()
    full.setup.chip_x_max = full.setup.chip_y_max = 100.0
    full.setup.telescope_turned_y_deg = 0.1
    run("rot", full, L.CF_IGNORE_DET_WINDOW | L.CF_IGNORE_GAS_ABS | L.CF_IGNORE_CONV_PROB)
''' % ROOT
for rep in range(int(os.environ.get("SART_AB_ROUNDS", "1"))):
    for lib in sys.argv[1:]:
        env = dict(os.environ)
        if lib != "default":
            env["SART_LIBSART"] = os.path.abspath(lib)
        print(lib, flush=True)
        subprocess.run([sys.executable, "-c", CHILD], env=env, check=True)
