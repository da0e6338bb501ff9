#!/usr/bin/env python3
"""Per-stage cycle breakdown of the ray kernel (diagnostic build: make -C solaraxionraytracing_amd/csrc clean all STAGE_TIMING=1).
Prints, per workload, the share of a wave's lifetime spent in stage A0 / A1 / B and the cycles per pass of each stage."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import solaraxionraytracing_amd as sa
from solaraxionraytracing_amd import _lib as L
from bench import make_setup

n = 100_000_000
SMALL = dict(n_radii=400, n_energies=300, refl_n_angles=200, refl_n_energies=200)   # tables that stay in every XCD's L2
for wl in ("babyiaxo_xmm", "cast_llnl_gold", "babyiaxo_xmm/small-tables"):
    if wl.endswith("small-tables"):
        full, flags = sa.initFullSetup(**SMALL), 0
    else:
        full, flags = make_setup(wl)
    with sa.RayTracer(full) as rt:
        rt.trace_histogram(n // 10, seed=1, flags=flags)
        rt.enable_kernel_timing(True)
        p = rt.trace_params(n, seed=1, flags=flags)
        import ctypes as C, numpy as np
        img = np.empty((256, 256)); summ = L.Summary()
        L.check(rt.lib.sart_trace_histogram(rt.handle, C.byref(p), L.as_dp(img), C.byref(summ)))
        ms, nl = rt.kernel_timing()
        v = list(summ.v)
    waves = 256 * 16
    a0, a1, b, tot = v[12], v[13], v[14], v[15]
    passes = n / 64.0
    n_a1 = (v[L.ACC["N_REACHED_TELESCOPE"]] if False else None)
    print("   in-kernel shader clock (s_memtime / s_memrealtime x 100 MHz, mean over waves): %.3f GHz" % (tot / max(v[L.ACC["N_HIT_NICKEL"]], 1.0) * 0.1))
    print("%s: kernel %.3f ms; wave lifetime %.0f cycles (%.2f GHz); share A0 %.3f A1 %.3f B %.3f other %.3f" % (
        wl, ms / nl, tot / waves, tot / waves / (ms / nl * 1e-3) / 1e9, a0 / tot, a1 / tot, b / tot, 1 - (a0 + a1 + b) / tot))
    print("   cycles per launched pass of 64 rays (per wave): A0 %.0f  A1 %.0f  B %.0f  total %.0f" % (
        a0 / passes, a1 / passes, b / passes, tot / passes))
    bs = [v[L.ACC[k]] for k in ("SUM_X", "SUM_Y", "SUM_R", "SUM_WEIGHTS_SQ", "N_OUTSIDE_IMAGE")]
    nb = v[L.ACC["N_SHELL_SELECTED"]] / 64.0
    print("   B sub-stages, cycles per B pass: mirror1+nickel %.0f | mirror2+detector plane %.0f | (to weights) %.0f -> energy index %.0f | "
          "energy row+reflectivity %.0f | window+strips %.0f" % (bs[0] / nb, bs[1] / nb, 0, bs[2] / nb, bs[3] / nb, bs[4] / nb))
    sel = v[L.ACC["N_SHELL_SELECTED"]] / 64.0
    print("   B passes %.3g -> %.0f cycles per B pass; shell-selected fraction %.3f" % (sel, b / max(sel, 1), sel / passes))
