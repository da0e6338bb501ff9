#!/usr/bin/env python3
"""Kernel time of the fused angular scan against the number of angles per scan (BabyIAXO / XMM, effective-area flags, chip 100 mm,
angles 0 .. 0.3 deg):  python tools/ascan_rate.py [rays]   ->  ps per (ray, angle) for 1 ... 64 angles (32 per launch)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import solaraxionraytracing_amd as sa
from solaraxionraytracing_amd import _lib as L

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200_000_000
full = sa.initFullSetup()
full.setup.chip_x_max = full.setup.chip_y_max = 100.0
flags = L.CF_IGNORE_DET_WINDOW | L.CF_IGNORE_GAS_ABS | L.CF_IGNORE_CONV_PROB
with sa.RayTracer(full) as rt:
    rt.trace_angular_scan(np.linspace(0, 0.3, 16), 20_000_000, seed=2, flags=flags)
    for k in (1, 2, 4, 8, 16, 24, 32, 50, 64):
        an = np.linspace(0.0, 0.3, k)
        rt.enable_kernel_timing(True)
        pa, sh = rt.trace_angular_scan(an, n, seed=1, flags=flags)
        ms, nl = rt.kernel_timing()
        rt.enable_kernel_timing(False)
        print("%3d angles  %2d launch(es)  %8.3f ms  %6.3f ps per (ray, angle)  %.3e (ray, angle)/s" % (k, nl, ms, ms * 1e9 / (n * k), n * k / (ms / 1e3)), flush=True)
