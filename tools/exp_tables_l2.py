#!/usr/bin/env python3
"""Experiment: the headline workload with the default tables (62 MB, L2 misses) and with tables small enough to stay in
every XCD's 4 MB L2 - how much of the kernel time is memory latency beyond L2?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import solaraxionraytracing_amd as sa
from solaraxionraytracing_amd import _lib as L

def run(name, full, n=100_000_000, reps=5):
    with sa.RayTracer(full) as rt:
        rt.trace_histogram(n // 10, seed=1)
        rt.enable_kernel_timing(True)
        for k in range(reps):
            img, s = rt.trace_histogram(n, seed=1, ray_id_offset=k * n, accumulate=(k > 0))
        ms, nl = rt.kernel_timing()
    print("%-60s %.3f ms / 1e8  %.3g rays/s  passed %.4f" % (name, ms / nl, n / (ms / nl) * 1e3, s["N_PASSED"] / s["N_RAYS"]), flush=True)

for rep in range(2):
    run("BabyIAXO default tables (1968x1500, 1000x1000)", sa.initFullSetup())
    run("BabyIAXO small tables (400x300, 200x200)", sa.initFullSetup(n_radii=400, n_energies=300, refl_n_angles=200, refl_n_energies=200))
    run("BabyIAXO tiny tables (100x100, 50x50)", sa.initFullSetup(n_radii=100, n_energies=100, refl_n_angles=50, refl_n_energies=50))
    run("CAST default", sa.initFullSetup(L.ES_CAST, L.DK_INGRID2018, L.SK_VACUUM, L.TK_LLNL, reflectivity="gold"))
    run("CAST small", sa.initFullSetup(L.ES_CAST, L.DK_INGRID2018, L.SK_VACUUM, L.TK_LLNL, reflectivity="gold", n_radii=400, n_energies=300, refl_n_angles=200, refl_n_energies=200))
