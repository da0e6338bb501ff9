#!/usr/bin/env python3
"""A long SART_ACCUM_FIXED64 accumulation on the headline workload, 1e9 rays per launch, reporting every 200 launches how far the
brightest pixel has come (profiles/*_fixed64_long_run.txt): where the status check starts to fail at a given headroom, and that
with the roll-over limbs (ROLLOVER=1: sart_rollover_accumulator_device behind every launch) it never does.

  HEADROOM=0|31 STEPS=2600 ROLLOVER=0|1 python tools/exp_fixed64_long.py

Scalars are printed by name from _lib.ACC / _lib.ACC_HI (round 4's version carried a hand-written name list in another order:
its scalar columns were mislabelled)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import solaraxionraytracing_amd as sa
from solaraxionraytracing_amd import _lib as L

full = sa.initFullSetup()
n = 1_000_000_000
roll = os.environ.get("ROLLOVER", "0") == "1"
N_IMG = 256 * 256
with sa.RayTracer(full) as rt:
    rt.set_accumulation_mode("fixed64", int(os.environ.get("HEADROOM", "0")))
    acc = torch.zeros(sa.accumulator_len(256), dtype=torch.int64, device="cuda:0")
    hi = torch.zeros_like(acc)
    out = torch.zeros(sa.accumulator_len(256), dtype=torch.float64, device="cuda:0")
    for k in range(int(os.environ.get("STEPS", "2600"))):
        p = rt.trace_params(n, seed=1, ray_id_offset=k * n, accumulate=(k > 0))
        rt.trace_histogram_device(p, acc.data_ptr())
        if roll:
            rt.rollover_accumulator_device(p, acc.data_ptr(), hi.data_ptr())
        if (k + 1) % 200 == 0:
            rt.finalize_accumulator_limbs_device(p, acc.data_ptr(), hi.data_ptr() if roll else None, out.data_ptr())
            try:
                rt.synchronize()
                ok = "ok"
            except Exception as e:
                ok = "FAIL: " + str(e)[:160]
            a, h = acc.cpu().numpy(), hi.cpu().numpy()
            value = h.astype(object) * 2 ** L.FIXED_LIMB_BITS + a.astype(object)          # exact integers
            img, sc = value[:N_IMG], value[N_IMG:N_IMG + L.SART_ACC_COUNT]
            two = lambda name: int(sc[L.ACC_HI[name]]) * 2 ** L.FIXED_LIMB_BITS + int(sc[L.ACC[name]])
            print(k + 1, ok, "brightest pixel 2^%.2f quanta" % np.log2(float(max(img))), "raw slot max 2^%.2f" % np.log2(float(a[:N_IMG].max())),
                  "min raw pixel", int(a[:N_IMG].min()),
                  {"N_RAYS": int(sc[L.ACC["N_RAYS"]]), "N_PASSED": int(sc[L.ACC["N_PASSED"]]), "SUM_WEIGHTS (quanta)": two("SUM_WEIGHTS"),
                   "SUM_WEIGHTS_SQ (quanta)": two("SUM_WEIGHTS_SQ"), "SUM_WEIGHTS_OUTSIDE lo / hi": (int(sc[17]), int(sc[18])),
                   "flux": float(out[N_IMG + L.ACC["SUM_WEIGHTS"]].item())}, flush=True)
