#!/usr/bin/env python3
"""bench.py — rays/second of the per-ray hot path on N MI355X GPUs of one node.

A "step" is one pass of the hot path (sample -> bore/pipes -> Wolter shells -> reflectivity -> detector ->
focal-plane histogram) over one batch of --rays-per-step rays per GPU, with all tables resident in HBM.
Workload = BASELINE.json configs[2]: BabyIAXO magnet + XMM-Newton shells (58), vacuum, InGridIAXO window,
256x256 focal-plane image; one step = one 1e9-ray image per GPU.  Inputs are the documented synthetic tables
(E1 Primakoff emission on AGSS09, G1 Henke gold reflectivity; the reference's own input files are not shipped).

Multi-GPU (weak scaling): one process per GPU (torchrun contract), rays shard by global ray id, every rank
accumulates its own image and ONE RCCL reduce of the fused accumulator closes the timed region.

Prints one JSON line on rank 0.  --profile-run skips the CPU baseline (for use under rocprofv3).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# algorithmic bytes per ray of the reference's formulation (f64 tables, no cache credit)
BYTES_KILLED, BYTES_MIRROR, BYTES_DETECTOR = 184.0, 184.0 + 64.0, 456.0
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
F64_VALU_PEAK_TFLOPS = 78.6    # 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz
FLOPS_DETECTOR, FLOPS_KILLED = 2000.0, 300.0   # SURVEY.md 8(d) secondary figure


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rays-per-step", type=float, default=1e9, help="rays per step and GPU (one BabyIAXO image of BASELINE configs[2])")
    ap.add_argument("--workload", default="babyiaxo_xmm", choices=["babyiaxo_xmm", "cast_llnl_gold"])
    ap.add_argument("--cpu-sample", type=float, default=6e8, help="rays of the CPU-baseline sample")
    ap.add_argument("--profile-run", action="store_true", help="no CPU baseline (run under rocprofv3)")
    ap.add_argument("--traffic-bytes-per-launch", type=float, default=None,
                    help="fabric bytes per launch from a separate rocprofv3 --pmc pass; default: profiles/pmc_traffic.json "
                         "if it was measured for the same workload and launch size")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    import solaraxionraytracing_amd as sa
    from solaraxionraytracing_amd import _lib as L, distributed as D

    # SART_BENCH_BACKEND=gloo + SART_BENCH_DEVICE=0: rehearsal of the multi-rank path on a one-GPU box
    rank, world, local_rank = D.init_process_group_from_env(os.environ.get("SART_BENCH_BACKEND"))
    if "SART_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["SART_BENCH_DEVICE"])
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    if args.workload == "babyiaxo_xmm":
        full = sa.initFullSetup()
        wl_name = "BabyIAXO magnet + XMM-Newton 58 shells, vacuum, InGridIAXO, 256x256 focal-plane image (BASELINE configs[2])"
    else:
        full = sa.initFullSetup(L.ES_CAST, L.DK_INGRID2018, L.SK_VACUUM, L.TK_LLNL, reflectivity="gold")
        wl_name = "CAST magnet + LLNL 14 shells, gold_0.25microns reflectivities (BASELINE configs[1])"

    rays = int(args.rays_per_step)
    traffic = args.traffic_bytes_per_launch
    traffic_note = None
    if traffic is None:
        try:   # PMC counters cannot be collected from inside this process; use the committed separate-pass measurement
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                t = json.load(f)
            if t["workload"] == args.workload:
                traffic = t["bytes_per_ray"] * rays
                traffic_note = t["note"]
        except Exception:
            pass
    rt = sa.RayTracer(full, device=local_rank)
    # Launches, torch ops on the accumulator and the RCCL reduce are ordered by ONE explicit stream.  (torch's
    # default stream has handle 0, which sart_set_stream reads as "the context's own stream".)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    rt.set_stream(stream.cuda_stream)
    acc = torch.zeros(sa.accumulator_len(256), dtype=torch.float64, device=dev)
    seed = 299792458

    def step(k: int):
        # global ray ids: step-major, rank-minor => the union over ranks and steps is a contiguous id range
        offset = (k * world + rank) * rays
        p = rt.trace_params(rays, seed=seed, ray_id_offset=offset, accumulate=True)
        rt.trace_histogram_device(p, acc.data_ptr())

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for k in range(args.warmup):
        step(10_000 + k)   # ray ids outside the timed range
    if world > 1:  # warm the communicator
        D.reduce_accumulator(acc.clone(), dst=0)
    barrier()
    acc.zero_()
    rt.enable_kernel_timing(True)
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    D.reduce_accumulator(acc, dst=0)     # the single RCCL reduce of the output histograms
    barrier()
    t1 = time.perf_counter()
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed_s = float(elapsed.item())
    kernel_ms, n_launch = rt.kernel_timing()
    rt.enable_kernel_timing(False)

    if rank == 0:
        host = acc.cpu().numpy()
        n_img = 256 * 256
        summ = {k: float(host[n_img + i]) for k, i in L.ACC.items()}
        total_rays = float(world) * args.steps * rays
        assert summ["N_RAYS"] == total_rays, (summ["N_RAYS"], total_rays)
        value = total_rays / elapsed_s
        # roofline of the dominant (only) kernel, per launch on this rank: algorithmic bytes of SURVEY 8(d)
        frac_det = summ["N_PASSED_TILL_WINDOW"] / total_rays
        frac_mirror = summ["N_SHELL_SELECTED"] / total_rays - frac_det
        frac_killed = 1.0 - summ["N_SHELL_SELECTED"] / total_rays
        bytes_per_ray = frac_killed * BYTES_KILLED + frac_mirror * BYTES_MIRROR + frac_det * BYTES_DETECTOR
        avg_kernel_s = kernel_ms / 1e3 / max(1, n_launch)
        achieved_gbs = bytes_per_ray * rays / avg_kernel_s / 1e9
        flops_per_ray = frac_killed * FLOPS_KILLED + (1.0 - frac_killed) * FLOPS_DETECTOR
        out = {
            "metric": "rays/sec",
            "value": value,
            "unit": "rays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed_s / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": wl_name, "rays_per_step_per_gpu": rays, "total_rays": total_rays,
                       "tables": full.meta, "sharding": "global ray id, 1 RCCL reduce of image+scalars",
                       "device": rt.device_info()},
            "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note,
                         "kernel": "trace_histogram_kernel", "avg_kernel_ms": avg_kernel_s * 1e3, "launches": n_launch,
                         "algorithmic_bytes_per_ray": bytes_per_ray, "headline_bytes_per_ray_upper_bound": BYTES_DETECTOR,
                         "secondary_f64_valu": {"achieved_tflops": flops_per_ray * rays / avg_kernel_s / 1e12,
                                                "peak_tflops": F64_VALU_PEAK_TFLOPS}},
            "results": {"flux": summ["SUM_WEIGHTS"], "passed_fraction": summ["N_PASSED"] / total_rays,
                        "reached_telescope_fraction": summ["N_REACHED_TELESCOPE"] / total_rays,
                        "shell_selected_fraction": summ["N_SHELL_SELECTED"] / total_rays},
        }
        if world == 1 and not args.profile_run:
            out["cpu_baseline"] = cpu_baseline(full, int(args.cpu_sample), seed)
            if args.workload == "babyiaxo_xmm":
                out["other_workloads"] = [other_workload_rate()]
                out["effective_area_rms"] = effective_area_rms()
        print(json.dumps(out))
    rt.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def other_workload_rate():
    """BASELINE configs[1] (CAST magnet + LLNL telescope, gold reflectivities, 1e8 rays) beside the headline workload:
    three 1e8-ray launches, HIP-event kernel time.  Informational; `value` is the BabyIAXO workload."""
    import solaraxionraytracing_amd as sa
    from solaraxionraytracing_amd import _lib as L
    full = sa.initFullSetup(L.ES_CAST, L.DK_INGRID2018, L.SK_VACUUM, L.TK_LLNL, reflectivity="gold")
    n = 100_000_000
    with sa.RayTracer(full) as rt:
        rt.trace_histogram(n // 10, seed=1)
        rt.enable_kernel_timing(True)
        for k in range(3):
            _, s = rt.trace_histogram(n, seed=1, ray_id_offset=k * n, accumulate=(k > 0))
        ms, n_launch = rt.kernel_timing()
    return {"workload": "CAST magnet + LLNL 14 shells, gold_0.25microns reflectivities, 1e8 rays (BASELINE configs[1])",
            "rays_per_s": n / (ms / n_launch) * 1e3, "ms_per_launch": ms / n_launch, "passed_fraction": s["N_PASSED"] / s["N_RAYS"]}


def effective_area_rms(points: int = 8, rays_per_angle: int = 1_000_000):
    """BASELINE's second metric (configs[3]): RMS between the max-normalised angular-scan curves (raytracer.nim:2791-2802) of
    the HIP path and of the CPU oracle for the same seed family; XMM shells, telescope_turned_y 0 .. 0.3 deg, effective-area
    flags, chip 100 mm.  tools/effarea_rms.py is the stand-alone version (profiles/r01_effective_area_rms.json)."""
    import numpy as np
    import solaraxionraytracing_amd as sa
    from solaraxionraytracing_amd import _lib as L
    from oracle.oracle import Oracle
    full = sa.initFullSetup()
    full.setup.chip_x_max = full.setup.chip_y_max = 100.0
    flags = L.CF_IGNORE_DET_WINDOW | L.CF_IGNORE_GAS_ABS | L.CF_IGNORE_CONV_PROB
    angles = np.linspace(0.0, 0.3, points)
    with sa.RayTracer(full) as rt:
        _, _, gpu_rel = sa.performAngularScan(rt, 0, 0, 1, rays_per_angle, flags=flags, angles=angles)
    o = Oracle(full)
    cpu = np.empty(points)
    for i, a in enumerate(angles):
        s = full.setup.copy()
        s.telescope_turned_y_deg = float(a)
        _, summ, _ = o.trace_histogram(rays_per_angle, ray_id_offset=i * rays_per_angle, flags=flags, setup=s,
                                       n_threads=available_cpus())
        cpu[i] = summ["SUM_WEIGHTS"]
    cpu_rel = cpu / cpu.max()
    return {"value": float(np.sqrt(np.mean((gpu_rel - cpu_rel) ** 2))), "points": points, "rays_per_angle": rays_per_angle,
            "gpu_relative_flux": [round(float(x), 6) for x in gpu_rel], "cpu_relative_flux": [round(float(x), 6) for x in cpu_rel]}


def available_cpus() -> int:
    """CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota (the GPU boxes give a
    container a share of the host's hardware threads)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                      # cgroup v2: "<quota> <period>" or "max <period>"
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:     # cgroup v1
                quota = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                period = int(f.read())
            if quota > 0:
                n = min(n, max(1, int(quota / period + 0.5)))
        except Exception:
            pass
    return n


def cpu_baseline(full, n_sample: int, seed: int):
    """The CPU oracle (restatement of the Nim path; the Nim binary cannot be built) timed on this host's cores on a
    bounded sample of the same workload."""
    from oracle.oracle import Oracle
    cores = available_cpus()
    o = Oracle(full)
    o.trace_histogram(200_000, seed=seed, n_threads=cores)   # warm up threads / page in tables
    t0 = time.perf_counter()
    _, summ, used = o.trace_histogram(n_sample, seed=seed, n_threads=cores)
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    n1 = max(200_000, n_sample // 50)
    o.trace_histogram(n1, seed=seed, n_threads=1)
    dt1 = time.perf_counter() - t0
    return {"value": n_sample / dt, "unit": "rays/s", "cores": used, "kind": "port",
            "sample": "%d rays of the same workload, C restatement of traceAxion (oracle/sart_oracle.c, gcc -O2 -fopenmp), "
                      "%.1f s; single thread: %.3g rays/s" % (n_sample, dt, n1 / dt1)}


if __name__ == "__main__":
    main()
