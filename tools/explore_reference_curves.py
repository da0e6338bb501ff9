#!/usr/bin/env python3
"""Exploration (GPU box): angular scan on the McXtrace grid and the CAST/LLNL parallel-beam effective area vs energy,
printed next to the curves the reference ships.  Feeds the bounds of tests/test_reference_data.py."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import solaraxionraytracing_amd as sa
from solaraxionraytracing_amd import _lib as L, tables

ref = tables.reference_curves()
ang = ref["mcxtrace_angle_deg"]
full = sa.initFullSetup()
full.setup.chip_x_max = full.setup.chip_y_max = 100.0
flags = L.CF_IGNORE_DET_WINDOW | L.CF_IGNORE_GAS_ABS | L.CF_IGNORE_CONV_PROB
with sa.RayTracer(full) as rt:
    _, flux, rel = sa.performAngularScan(rt, 0, 0, 1, 4_000_000, flags=flags, angles=ang)
ok = ref["xmm_effective_area"] > 0
xmm_x = ref["xmm_angle_arcmin"] / 60.0
xmm = np.interp(ang, xmm_x[ok], ref["xmm_effective_area"][ok] / ref["xmm_effective_area"].max())
print("angle  gpu  mcxtrace  xmm_theory")
for a, g, m, x in zip(ang, rel, ref["mcxtrace_rel_flux"], xmm):
    print("%.3f  %.4f  %.4f  %.4f" % (a, g, m, x))

# CAST / LLNL, parallel beam through the bore
d = tables.llnl_effective_area() if hasattr(tables, "llnl_effective_area") else None
area_cm2 = np.pi * 2.15 ** 2
fl = L.CF_XRAY_TEST | L.CF_IGNORE_DET_WINDOW | L.CF_IGNORE_GAS_ABS | L.CF_IGNORE_CONV_PROB
print("E  area_llnl_4coat  area_gold  thesis")
for energy in (0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0, 8.0, 9.0):
    out = []
    for refl in ("henke", "gold"):
        src = L.TestSourceConfig()
        src.active, src.parallel = 1, 1
        src.energy, src.distance, src.radius, src.activity = energy, 100.0, 21.5, 1.0
        src.offAxisUp = src.offAxisLeft = 0.0
        src.lengthCol = 0.0
        full = sa.initFullSetup(L.ES_CAST, L.DK_INGRID2018, L.SK_VACUUM, L.TK_LLNL, flags=fl, source_cfg=src, reflectivity=refl)
        with sa.RayTracer(full) as rt:
            _, s = rt.trace_histogram(2_000_000, seed=4, flags=fl)
        out.append(area_cm2 * s["SUM_WEIGHTS"] / s["N_RAYS"])
        geo = s["N_PASSED"] / s["N_RAYS"]
    th = np.interp(energy, d[0], d[1]) if d is not None else float("nan")
    print("%.1f  %.3f  %.3f  %.3f   (geometric pass fraction %.3f)" % (energy, out[0], out[1], th, geo))
