"""Wall time of performAngularScan (50 angles) at three ray counts: what a change of the telescope angle costs on the host (re-hoisting,
zones, blob upload) beside the kernel.  python tools/angular_scan_time.py  (GPU box)"""
import time, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import solaraxionraytracing_amd as sa
from solaraxionraytracing_amd.raytracer import performAngularScan
full = sa.initFullSetup()
with sa.RayTracer(full) as rt:
    rt.trace_histogram(100000, seed=1)
    for n in (100_000, 10_000_000, 100_000_000):
        t0 = time.time()
        a, f, r = performAngularScan(rt, 0.0, 0.7, 50, n, seed=5)
        dt = time.time() - t0
        print("50 angles x %.0e rays: %.3f s wall = %.2f ms per angle (kernel share at 6.6e10 rays/s: %.2f ms)" % (n, dt, dt / 50 * 1e3, n / 6.6e10 * 1e3), flush=True)
    print(r[:5])
