#!/usr/bin/env python3
"""BASELINE's second metric: RMS between the max-normalised angular-scan (effective-area) curves of the HIP path and of
the CPU oracle for the same seed family (raytracer.nim:2791-2802; BASELINE configs[3]: XMM shells, telescope_turned_y scan,
flags ignoreDetWindow / ignoreGasAbs / ignoreConvProb, chip 100 mm).  Prints one JSON line."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import solaraxionraytracing_amd as sa
from solaraxionraytracing_amd import _lib as L
from oracle.oracle import Oracle

ap = argparse.ArgumentParser()
ap.add_argument("--points", type=int, default=16)
ap.add_argument("--max-angle", type=float, default=0.3)
ap.add_argument("--rays", type=float, default=2e6, help="rays per angle")
args = ap.parse_args()
n = int(args.rays)
full = sa.initFullSetup()
full.setup.chip_x_max = full.setup.chip_y_max = 100.0
flags = L.CF_IGNORE_DET_WINDOW | L.CF_IGNORE_GAS_ABS | L.CF_IGNORE_CONV_PROB
angles = np.linspace(0.0, args.max_angle, args.points)
t = time.perf_counter()
with sa.RayTracer(full) as rt:
    _, gpu, gpu_rel = sa.performAngularScan(rt, 0, 0, 1, n, flags=flags, angles=angles)
    _, gpu2, gpu2_rel = sa.performAngularScan(rt, 0, 0, 1, n, seed=12345, flags=flags, angles=angles)
t_gpu = time.perf_counter() - t
t = time.perf_counter()
o = Oracle(full)
cpu = np.empty_like(gpu)
for i, a in enumerate(angles):          # same ray-id blocks as the host driver: angle i uses ids [i n, (i + 1) n)
    s = full.setup.copy()
    s.telescope_turned_y_deg = float(a)
    _, summ, _ = o.trace_histogram(n, ray_id_offset=i * n, flags=flags, setup=s)
    cpu[i] = summ["SUM_WEIGHTS"]
t_cpu = time.perf_counter() - t
cpu_rel = cpu / cpu.max()
rms = float(np.sqrt(np.mean((gpu_rel - cpu_rel) ** 2)))
rms_other_seed = float(np.sqrt(np.mean((gpu2_rel - cpu_rel) ** 2)))
print(json.dumps({"metric": "effective-area curve RMS vs CPU ref", "value": rms, "points": args.points, "rays_per_angle": n,
                  "angles_deg": angles.round(6).tolist(), "gpu_relative_flux": gpu_rel.round(8).tolist(),
                  "cpu_relative_flux": cpu_rel.round(8).tolist(),
                  "rms_same_seed_family": rms, "rms_different_seed_family_(Monte_Carlo_error)": rms_other_seed,
                  "max_abs_flux_rel_diff_same_seed": float(np.max(np.abs(gpu / cpu - 1.0))),
                  "seconds": {"gpu_two_scans_incl_setup": t_gpu, "cpu_oracle_one_scan": t_cpu}}))
