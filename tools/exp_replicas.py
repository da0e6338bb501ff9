#!/usr/bin/env python3
"""Experiment: image replicas for the wide BabyIAXO image (SART_IMAGE_REPLICAS; with the DEBUG_KNOBS build
SART_DEBUG_FLAGS=08000000 keys the replica by workgroup = XCD instead of by wave)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import solaraxionraytracing_amd as sa

def run(full, env, n=100_000_000, reps=4):
    for k in ("SART_IMAGE_REPLICAS", "SART_DEBUG_FLAGS", "SART_DEBUG_NO_IMAGE_ATOMICS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    with sa.RayTracer(full) as rt:
        rt.trace_histogram(n // 10, seed=1)
        rt.enable_kernel_timing(True)
        for k in range(reps):
            img, s = rt.trace_histogram(n, seed=1, ray_id_offset=k * n, accumulate=(k > 0))
        ms, nl = rt.kernel_timing()
    print("%-64s %.3f ms / 1e8   flux %.6e" % (env, ms / nl, s["SUM_WEIGHTS"]), flush=True)

full = sa.initFullSetup()
run(full, {})
run(full, {"SART_DEBUG_NO_IMAGE_ATOMICS": "1"})
for r in ("4", "8", "16"):
    run(full, {"SART_IMAGE_REPLICAS": r})
    run(full, {"SART_IMAGE_REPLICAS": r, "SART_DEBUG_FLAGS": "08000000"})
run(full, {})
