#!/usr/bin/env python3
"""Experiment: kernel time against launch size (fixed cost per launch / tail of the persistent grid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import solaraxionraytracing_amd as sa
full = sa.initFullSetup()
with sa.RayTracer(full) as rt:
    rt.trace_histogram(100_000_000, seed=1)
    for n in (1_000_000, 10_000_000, 50_000_000, 100_000_000, 200_000_000, 500_000_000, 1_000_000_000, 2_000_000_000):
        reps = max(3, min(40, int(3e9 // n)))
        rt.enable_kernel_timing(True)
        for k in range(reps):
            rt.trace_histogram(n, seed=1, ray_id_offset=k * n, accumulate=(k > 0))
        ms, nl = rt.kernel_timing()
        rt.enable_kernel_timing(False)
        print("n = %.0e: %.4f ms per launch, %.4g rays/s  (%d launches)" % (n, ms / nl, n / (ms / nl) * 1e3, nl), flush=True)
