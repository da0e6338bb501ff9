#!/usr/bin/env python3
"""Experiment (DEBUG_KNOBS build, SART_LIBSART=tools/microbench/libsart_dbg.so): kernel time of the headline workload with
stages switched off - the marginal cost of each stage under the real contention of the others."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import solaraxionraytracing_amd as sa
from solaraxionraytracing_amd import _lib as L

def run(name, full, env, n=100_000_000, reps=4):
    for k in list(os.environ):
        if k.startswith("SART_DEBUG") or k in ("SART_NO_EARLY_REJECT",):
            del os.environ[k]
    os.environ.update(env)
    with sa.RayTracer(full) as rt:          # knobs are read when the context is created
        rt.trace_histogram(n // 10, seed=1)
        rt.enable_kernel_timing(True)
        for k in range(reps):
            img, s = rt.trace_histogram(n, seed=1, ray_id_offset=k * n, accumulate=(k > 0))
        ms, nl = rt.kernel_timing()
    print("%-14s %-44s %.3f ms / 1e8   shell %.3f passed %.3f" % (name, env, ms / nl, s["N_SHELL_SELECTED"] / s["N_RAYS"], s["N_PASSED"] / s["N_RAYS"]), flush=True)

setups = [("BabyIAXO", sa.initFullSetup()),
          ("CAST", sa.initFullSetup(L.ES_CAST, L.DK_INGRID2018, L.SK_VACUUM, L.TK_LLNL, reflectivity="gold"))]
for name, full in setups:
    for env in ({}, {"SART_DEBUG_NO_IMAGE_ATOMICS": "1"}, {"SART_DEBUG_FLAGS": "10000000"}, {"SART_DEBUG_FLAGS": "20000000"},
                {"SART_NO_EARLY_REJECT": "1"}, {"SART_NO_EARLY_REJECT": "1", "SART_DEBUG_FLAGS": "10000000"}, {}):
        run(name, full, env)
