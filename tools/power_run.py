#!/usr/bin/env python3
"""Keeps the ray kernel running for --seconds (1e9-ray launches, no checks) so that tools/power_sample.sh can read socket
power and shader clock beside it; works with the DEBUG_KNOBS build and its stage-ablation flags."""
import sys, os, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import solaraxionraytracing_amd as sa
from bench import make_setup
ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=25.0)
ap.add_argument("--workload", default="babyiaxo_xmm")
a = ap.parse_args()
full, flags = make_setup(a.workload)
n = 1_000_000_000
with sa.RayTracer(full) as rt:
    rt.trace_histogram(n // 10, seed=1, flags=flags)
    rt.enable_kernel_timing(True)
    t0 = time.time(); k = 0
    while time.time() - t0 < a.seconds:
        rt.trace_histogram(n, seed=1, ray_id_offset=k * n, flags=flags, accumulate=(k > 0)); k += 1
    ms, nl = rt.kernel_timing()
print("%s %s: %.3f ms per 1e9-ray launch, %.4g rays/s" % (a.workload, {k: v for k, v in os.environ.items() if k.startswith("SART_") and k != "SART_LIBSART"}, ms / nl, n / (ms / nl) * 1e3))
