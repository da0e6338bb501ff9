#!/usr/bin/env python3
"""A/B timing of kernel builds: python tools/exp_ab.py lib1.so lib2.so ...  (each in its own subprocess, alternating)."""
import sys, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os
sys.path.insert(0, %r)
import solaraxionraytracing_amd as sa
from solaraxionraytracing_amd import _lib as L
def run(name, full, n=100_000_000, reps=5):
    with sa.RayTracer(full) as rt:
        rt.trace_histogram(n // 10, seed=1)
        rt.enable_kernel_timing(True)
        for k in range(reps):
            img, s = rt.trace_histogram(n, seed=1, ray_id_offset=k * n, accumulate=(k > 0))
        ms, nl = rt.kernel_timing()
    print("  %%-10s %%.3f ms / 1e8  flux %%.9e passed %%d" %% (name, ms / nl, s["SUM_WEIGHTS"], s["N_PASSED"]), flush=True)
run("BabyIAXO", sa.initFullSetup())
run("CAST", sa.initFullSetup(L.ES_CAST, L.DK_INGRID2018, L.SK_VACUUM, L.TK_LLNL, reflectivity="gold"))
run("gas", sa.initFullSetup(stage=L.SK_GAS))
''' % ROOT
for rep in range(1):
    for lib in sys.argv[1:]:
        env = dict(os.environ)
        if lib != "default":
            env["SART_LIBSART"] = os.path.abspath(lib)
        print(lib, flush=True)
        subprocess.run([sys.executable, "-c", CHILD], env=env, check=True)
