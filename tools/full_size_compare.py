#!/usr/bin/env python3
"""BASELINE configs[2] (or configs[1]: --workload cast_llnl_gold --rays 1e8) at full size on both sides: the 256 x 256 focal-plane image and the counters of N rays (default 1e9)
from the HIP path and from the CPU oracle (same seeds and ray ids), compared directly.  Prints one JSON line."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import solaraxionraytracing_amd as sa
from oracle.oracle import Oracle

ap = argparse.ArgumentParser()
ap.add_argument("--rays", type=float, default=1e9)
ap.add_argument("--chunk", type=float, default=2.5e8, help="rays per oracle call (progress lines in between)")
ap.add_argument("--workload", default="babyiaxo_xmm", choices=["babyiaxo_xmm", "cast_llnl_gold", "cast_llnl", "cast_abrixas", "babyiaxo_xmm_rot", "babyiaxo_xmm_gas"],
                help="BASELINE configs[2] / configs[1] / one angle bin of configs[3] (rotated, 100 mm chip, effective-area flags) / configs[4]'s gas stage")
args = ap.parse_args()
n, chunk = int(args.rays), int(args.chunk)
from solaraxionraytracing_amd import _lib as L
flags = None
if args.workload == "babyiaxo_xmm":
    full = sa.initFullSetup()
elif args.workload == "cast_llnl_gold":
    full = sa.initFullSetup(L.ES_CAST, L.DK_INGRID2018, L.SK_VACUUM, L.TK_LLNL, reflectivity="gold")
elif args.workload == "cast_llnl":          # the reference's own LLNL pairing: four multilayer coatings by shell group (raytracer.nim:1164-1187)
    full = sa.initFullSetup(L.ES_CAST, L.DK_INGRID2018, L.SK_VACUUM, L.TK_LLNL)
elif args.workload == "cast_abrixas":       # raytracer.nim:1320-1346
    full = sa.initFullSetup(L.ES_CAST, L.DK_INGRID2017, L.SK_VACUUM, L.TK_ABRIXAS)
elif args.workload == "babyiaxo_xmm_gas":
    full = sa.initFullSetup(stage=L.SK_GAS)
else:
    full = sa.initFullSetup()
    full.setup.telescope_turned_x_deg, full.setup.telescope_turned_y_deg = 0.02, 0.1
    full.setup.chip_x_max = full.setup.chip_y_max = 100.0
    full.flags = flags = L.CF_IGNORE_DET_WINDOW | L.CF_IGNORE_GAS_ABS | L.CF_IGNORE_CONV_PROB
t = time.perf_counter()
with sa.RayTracer(full) as rt:
    img, s = rt.trace_histogram(n, seed=299792458, flags=flags)
t_gpu = time.perf_counter() - t
o = Oracle(full)
oimg = np.zeros_like(img)
osum = {}
t = time.perf_counter()
for off in range(0, n, chunk):
    m = min(chunk, n - off)
    im, sm, _ = o.trace_histogram(m, seed=299792458, ray_id_offset=off, flags=flags, n_threads=len(os.sched_getaffinity(0)))
    oimg += im
    for k, v in sm.items():
        osum[k] = osum.get(k, 0.0) + v
    print("oracle: %d / %d rays, %.0f s" % (off + m, n, time.perf_counter() - t), file=sys.stderr, flush=True)
t_cpu = time.perf_counter() - t
keys = ("N_RAYS", "N_REACHED_TELESCOPE", "N_SHELL_SELECTED", "N_HIT_NICKEL", "N_PASSED_TILL_WINDOW", "N_PASSED")
lit = oimg > 0.01 * oimg.max()
coarse = lambda a: a.reshape(32, 8, 32, 8).sum(axis=(1, 3))
out = {"workload": args.workload, "rays": n, "seconds": {"gpu_call_incl_copies": t_gpu, "cpu_oracle": t_cpu},
       "counters_gpu": {k: s[k] for k in keys}, "counters_oracle": {k: osum[k] for k in keys},
       "counter_differences": {k: s[k] - osum[k] for k in keys},
       "sum_weights_rel_diff": s["SUM_WEIGHTS"] / osum["SUM_WEIGHTS"] - 1.0,
       "image_l1_rel_diff": float(np.abs(img - oimg).sum() / oimg.sum()),
       "image_max_pixel_rel_diff_where_lit": float(np.max(np.abs(img[lit] / oimg[lit] - 1.0))),
       "image_8x8_blocks_max_rel_diff": float(np.max(np.abs(coarse(img) - coarse(oimg)) / coarse(oimg).max())),
       "mean_rays_per_lit_pixel": float(osum["N_PASSED"] / lit.sum()),
       "note": "differences are rays within the f64 oracle's own rounding noise (~1e-3 mm in the focal plane) of a cut or pixel edge"}
print(json.dumps(out))
