#!/usr/bin/env python3
"""A/B of the stage-A0 zones for turned telescopes (round 6; sart_api.hip: build_zones with the tilt's margin) against the state
before (SART_NO_TILT_ZONES: a turned telescope keeps the pipes' zone only), alternating subprocesses on one box:
the rotated single launch (one bin of BASELINE configs[3]) and the fused angular scan at 16 and 32 angles.  Results must be
the same to the last bit (FIXED64 sums and counters are printed).   python tools/exp_tilt_zones.py [rounds]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os
sys.path.insert(0, %r)
import numpy as np
import solaraxionraytracing_amd as sa
from solaraxionraytracing_amd import _lib as L
flags = L.CF_IGNORE_DET_WINDOW | L.CF_IGNORE_GAS_ABS | L.CF_IGNORE_CONV_PROB
full = sa.initFullSetup()
full.setup.chip_x_max = full.setup.chip_y_max = 100.0
full.setup.telescope_turned_y_deg = 0.1
N = 100_000_000
with sa.RayTracer(full) as rt:
    rt.set_accumulation_mode("fixed64")
    for k in range(3):
        rt.trace_flux(N, seed=2, ray_id_offset=k * N, flags=flags)
    rt.enable_kernel_timing(True)
    for k in range(5):
        img, s = rt.trace_histogram(N, seed=1, ray_id_offset=k * N, accumulate=(k > 0), flags=flags)
    ms, nl = rt.kernel_timing()
    print("  rot 0.1 deg   %%.4f ms / 1e8 rays   flux %%s passed %%d reached %%d" %% (ms / nl, float(s["SUM_WEIGHTS"]).hex(), s["N_PASSED"], s["N_REACHED_TELESCOPE"]), flush=True)
    rt.set_telescope_angles(turned_y_deg=0.0)
    M = 200_000_000
    for K in (16, 32):
        an = np.linspace(0.0, 0.3, K)
        rt.trace_angular_scan(an, 20_000_000, seed=2, flags=flags)
        rt.kernel_timing()
        pa, sh = rt.trace_angular_scan(an, M, seed=1, flags=flags)
        ms, nl = rt.kernel_timing()
        print("  ascan %%2d      %%.3f ms / 2e8 rays  %%.3f ps per (ray, angle)  flux[-1] %%s passed %%s reached %%d" %% (
            K, ms, ms * 1e9 / (M * K), float(pa["SUM_WEIGHTS"][-1]).hex(), int(pa["N_PASSED"].sum()), sh["N_REACHED_TELESCOPE"]), flush=True)
''' % ROOT
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    for knob in (False, True):
        env = {k: v for k, v in os.environ.items() if k != "SART_NO_TILT_ZONES"}
        if knob:
            env["SART_NO_TILT_ZONES"] = "1"
        print("SART_NO_TILT_ZONES=1 (before round 6)" if knob else "tilt zones (round 6)", flush=True)
        subprocess.run([sys.executable, "-c", CHILD], env=env, check=True)
