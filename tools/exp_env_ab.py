#!/usr/bin/env python3
"""A/B timing of environment knobs of libsart (read when a context is created): python tools/exp_env_ab.py KNOB[=VALUE] ..."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import solaraxionraytracing_amd as sa
from solaraxionraytracing_amd import _lib as L

def run(name, full, env, n=100_000_000, reps=5):
    for k in list(os.environ):
        if k.startswith("SART_") and k != "SART_LIBSART":
            del os.environ[k]
    os.environ.update(env)
    with sa.RayTracer(full) as rt:
        rt.trace_histogram(n // 10, seed=1)
        rt.enable_kernel_timing(True)
        for k in range(reps):
            img, s = rt.trace_histogram(n, seed=1, ray_id_offset=k * n, accumulate=(k > 0))
        ms, nl = rt.kernel_timing()
    print("%-10s %-34s %.3f ms / 1e8  flux %.12e passed %d img %.12e" % (name, env, ms / nl, s["SUM_WEIGHTS"], s["N_PASSED"], img.sum()), flush=True)

knobs = [dict([a.split("=", 1) if "=" in a else (a, "1")]) for a in sys.argv[1:]]
setups = [("BabyIAXO", sa.initFullSetup()), ("CAST", sa.initFullSetup(L.ES_CAST, L.DK_INGRID2018, L.SK_VACUUM, L.TK_LLNL, reflectivity="gold"))]
for rep in range(2):
    for name, full in setups:
        for env in [{}] + knobs:
            run(name, full, env)
