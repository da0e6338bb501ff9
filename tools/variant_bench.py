#!/usr/bin/env python3
"""Kernel-tuning helper (GPU box): times the BabyIAXO and CAST workloads for every build under
solaraxionraytracing_amd/variants/ (one subprocess per build: the library is chosen at import through SART_LIBSART)."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys; sys.path.insert(0, %r)
import solaraxionraytracing_amd as sa
from solaraxionraytracing_amd import _lib as L
def run(full, n=100_000_000, reps=6):
    with sa.RayTracer(full) as rt:
        rt.trace_histogram(n // 10, seed=1)
        rt.enable_kernel_timing(True)
        for k in range(reps):
            rt.trace_histogram(n, seed=1, ray_id_offset=k * n, accumulate=(k > 0))
        ms, nl = rt.kernel_timing()
    return ms / nl
a = run(sa.initFullSetup())
b = run(sa.initFullSetup(L.ES_CAST, L.DK_INGRID2018, L.SK_VACUUM, L.TK_LLNL, reflectivity="gold"))
c = run(sa.initFullSetup(stage=L.SK_GAS))
print("%%.3f %%.3f %%.3f" %% (a, b, c))
''' % ROOT
for lib in sorted(glob.glob(os.path.join(ROOT, "solaraxionraytracing_amd", "variants", "libsart_*.so"))):
    env = dict(os.environ, SART_LIBSART=lib)
    out = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    print(os.path.basename(lib), out.stdout.strip() or out.stderr[-300:], flush=True)
