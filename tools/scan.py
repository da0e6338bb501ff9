#!/usr/bin/env python3
"""Scan drivers of the hot path, one process per GPU:

  angular   performAngularScan (raytracer.nim:2778-2815; CLI --angularScanMin/Max --numAngularScanPoints, :2817-2862):
            effective-area style scan of telescope_turned_y, angle bins sharded over the ranks (BASELINE configs[3]).
            Writes a CSV (the reference only makes a PDF) and compares with the two curves the reference overlays
            (xmm_newton_angular_effective_area.csv, McXtrace_angular_xmm.csv; :2805-2813).
            Default: the reference's shape, a flux-only re-trace per angle on fresh rays.  --fused: the FUSED scan kernel
            (sart_trace_angular_scan): every ray is sampled and taken through bore and pipes once and turned through every
            angle (--shard rays, the default with --fused: every rank traces its share of the ray ids for all angles, one
            reduce of the scan accumulator - the cheaper sharding by the kernel's own cost model; --shard bins:
            the angles are dealt out to the ranks, every rank traces all ray ids for its angles) - all angles see the same
            rays, the curve does not depend on the number of ranks.
  mass      gas-stage axion-mass scan on the full AGSS09 emission table (BASELINE configs[4]) through the FUSED scan kernel
            (sart_trace_mass_scan): every rank traces its share of the ray ids ONCE and weighs each ray for every mass; one
            reduce of the scan accumulator (8 (points + 1) slots) over the ranks closes the scan ("1e10 rays across 8 MI355X
            with RCCL histogram reduce": --points 32 --rays 1e10 on 8 ranks).  All masses see the same rays (common random
            numbers).  --host-loop: the reference-shaped scan instead (a re-trace per mass point; --shard bins | rays).

Examples
  python tools/scan.py angular --angularScanMin 0 --angularScanMax 0.3 --numAngularScanPoints 16 --rays 1e7
  python tools/scan.py mass --gpus 8 --points 32 --rays 3e8          (starts its own 8 ranks; RCCL)
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/scan.py mass --points 32 --rays 3e8
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["angular", "mass"])
    ap.add_argument("--angularScanMin", type=float, default=0.0)
    ap.add_argument("--angularScanMax", type=float, default=0.3)
    ap.add_argument("--numAngularScanPoints", type=int, default=16)
    ap.add_argument("--points", type=int, default=32, help="mass scan points")
    ap.add_argument("--massMin", type=float, default=0.0)
    ap.add_argument("--massMax", type=float, default=0.02, help="eV; the literal-units gas stage has m_gamma = 0.008235 eV")
    ap.add_argument("--rays", type=float, default=1e7, help="rays per scan point")
    ap.add_argument("--chip", type=float, default=100.0, help="chip size in mm for the angular scan (SURVEY App. C)")
    ap.add_argument("--shard", default=None, choices=["bins", "rays"],
                    help="angular scan, mass --host-loop: bins = every scan point is a full run on one rank (BASELINE configs[3]'s "
                         "wording; the default of the re-trace scans); rays = every rank traces its share of the ray ids of every point and "
                         "the accumulators are reduced once.  Default with --fused: rays - the fused kernel pays its shared part "
                         "(sampling, bore, pipes, energy draw: ~12 ps per ray) once per launch and ~5.7 ps per (ray, angle), so G ranks "
                         "cost (L t_shared + K t_angle) / G per ray by ray id (L = ceil(K / 32) launches) against t_shared ceil(ceil(K / G) / 32) + "
                         "ceil(K / G) t_angle by angle group: by-ray is never slower and 19 %% faster at K = 50, G = 8 (DESIGN.md 6).  The "
                         "fused mass scan always shards the rays")
    ap.add_argument("--xrayTest", action="store_true",
                    help="angular: the parallel X-ray test source in front of the bore (raytracer.nim:1765-1806) instead of the sun - "
                         "the cleaner effective-area probe (SURVEY 8(d) config 4)")
    ap.add_argument("--fused", action="store_true", help="angular: the fused scan kernel (same rays for every angle) instead of a re-trace per angle")
    ap.add_argument("--host-loop", action="store_true",
                    help="mass: one re-trace per mass point (what the fused scan replaces; independent ray blocks per point)")
    ap.add_argument("--emission", default=None, choices=["agss09-device", "agss09", "primakoff", "legacy", "flat"],
                    help="solar emission table; default: agss09-device (all terms of readOpacityFile.nim on the AGSS09 model; emission "
                         "kernel -> CDFs -> guide tables without leaving the GPU, sart_emission_to_solar_tables) for the mass scan = "
                         "BASELINE configs[4]; agss09 = the same table through the host; primakoff (E1) for the angular scan")
    ap.add_argument("--accumulation", default="f64", choices=["f64", "fixed64"],
                    help="--shard rays only: fixed64 = deterministic integer accumulation + int64 reduce: the curve does not depend on "
                         "the number of ranks to the last bit (SART_ACCUM_FIXED64)")
    ap.add_argument("--out", default="gpurun_out/scan.csv")
    ap.add_argument("--gpus", type=int, default=None,
                    help="number of ranks (one per GPU).  Stand-alone: N > 1 starts N copies of this script, one rank each; "
                         "under torchrun it must equal WORLD_SIZE (default: WORLD_SIZE, or 1)")
    args = ap.parse_args()

    from solaraxionraytracing_amd import distributed as D
    n_ranks = args.gpus if args.gpus is not None else int(os.environ.get("WORLD_SIZE", "1"))
    rc = D.launch_ranks_if_needed(n_ranks, os.path.abspath(__file__), sys.argv[1:])   # exit code 2 on a mismatch, before any GPU call
    if rc is not None:
        raise SystemExit(rc)

    import torch
    import solaraxionraytracing_amd as sa
    from solaraxionraytracing_amd import _lib as L, tables

    rank, world, local_rank = D.init_process_group_from_env(os.environ.get("SART_BENCH_BACKEND"))
    if "SART_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["SART_BENCH_DEVICE"])
    if args.shard is None:
        args.shard = "rays" if (args.mode == "angular" and args.fused) else "bins"
    D.heartbeat("tables")
    n_rays = int(args.rays)
    emission = args.emission or ("agss09-device" if args.mode == "mass" else "primakoff")
    if args.mode == "angular":
        flags = L.CF_IGNORE_DET_WINDOW | L.CF_IGNORE_GAS_ABS | L.CF_IGNORE_CONV_PROB   # cf. comment :2315
        if args.xrayTest:
            flags |= L.CF_XRAY_TEST
        full = sa.initFullSetup(flags=flags if args.xrayTest else 0, emission=emission)
        full.setup.chip_x_max = full.setup.chip_y_max = args.chip       # ChipXMax = 100 mm alternative (raytracer.nim:262-264)
        xs = np.linspace(args.angularScanMin, args.angularScanMax, args.numAngularScanPoints)
    else:
        full = sa.initFullSetup(stage=L.SK_GAS, emission=emission)   # BASELINE configs[4]: full AGSS09 emission + m_a scan
        flags = 0
        xs = np.linspace(args.massMin, args.massMax, args.points)
    fused = args.mode == "mass" and not args.host_loop
    errs = None
    if args.mode == "angular" and args.fused and args.shard == "rays":
        # Fused angular scan, rays sharded: rank r turns ray ids [lo_r, hi_r) through ALL angles into a device scan accumulator;
        # ONE reduce of 8 (points + 1) slots over the ranks (int64 when fixed64).
        use_cuda = not (world > 1 and torch.distributed.get_backend() != "nccl")
        fixed64 = args.accumulation == "fixed64"
        acc = torch.zeros(sa.angular_scan_len(len(xs)), dtype=torch.float64, device=torch.device("cuda", local_rank))
        lo, hi = D.shard_range(n_rays, rank, world)
        with sa.RayTracer(full, device=local_rank) as rt:
            stream = torch.cuda.Stream(device=acc.device)
            torch.cuda.set_stream(stream)
            rt.set_stream(stream.cuda_stream)
            rt.set_accumulation_mode(args.accumulation)
            p = rt.trace_params(hi - lo, ray_id_offset=lo, flags=flags, accumulate=True)
            rt.trace_angular_scan_device(p, xs, acc.data_ptr())
            red = acc if use_cuda else acc.cpu()
            D.reduce_accumulator(red, dst=0, fixed64=fixed64)
            if fixed64:
                if not use_cuda:
                    acc.copy_(red)
                rt.finalize_angular_scan_device(p, len(xs), acc.data_ptr())
                rt.synchronize()
                red = acc
            red = red.cpu()
        per_angle, shared = sa.split_angular_scan(red.numpy(), len(xs))
        curve = per_angle["SUM_WEIGHTS"]
        errs = np.sqrt(per_angle["SUM_WEIGHTS_SQ"])
        if rank == 0:
            assert shared["N_RAYS"] == n_rays, (shared, n_rays)
        mine = None
    elif fused:
        # Fused scan: rank r traces ray ids [lo_r, hi_r) ONCE for all masses into a device scan accumulator; ONE reduce of
        # 8 (points + 1) slots over the ranks (int64 when fixed64: the curve then does not depend on the number of ranks).
        use_cuda = not (world > 1 and torch.distributed.get_backend() != "nccl")
        fixed64 = args.accumulation == "fixed64"
        acc = torch.zeros(sa.mass_scan_len(len(xs)), dtype=torch.float64, device=torch.device("cuda", local_rank))
        lo, hi = D.shard_range(n_rays, rank, world)
        with sa.RayTracer(full, device=local_rank) as rt:
            stream = torch.cuda.Stream(device=acc.device)
            torch.cuda.set_stream(stream)
            rt.set_stream(stream.cuda_stream)
            rt.set_accumulation_mode(args.accumulation)
            p = rt.trace_params(hi - lo, ray_id_offset=lo, flags=flags, accumulate=True)
            rt.trace_mass_scan_device(p, xs, acc.data_ptr())
            red = acc if use_cuda else acc.cpu()
            D.reduce_accumulator(red, dst=0, fixed64=fixed64)
            if fixed64:      # raw integers -> doubles (the quanta are a function of the inputs: the same on every rank)
                if not use_cuda:
                    acc.copy_(red)
                rt.finalize_mass_scan_device(p, xs, acc.data_ptr())
                rt.synchronize()   # raises if the integers did not resolve the weights / a slot wrapped
                red = acc
            red = red.cpu()
        per_mass, shared = sa.split_mass_scan(red.numpy(), len(xs))
        curve = per_mass["SUM_WEIGHTS"]
        errs = np.sqrt(per_mass["SUM_WEIGHTS_SQ"])
        if rank == 0:
            assert shared["N_RAYS"] == n_rays, (shared, n_rays)
        mine = None
    elif args.shard == "rays":
        # Every point: rank r traces ray ids [lo_r, hi_r) of the point's id block into a device accumulator, then ONE
        # reduce of the fused accumulator (image + scalars) over the ranks — the RCCL histogram reduce of config 5.
        use_cuda = not (world > 1 and torch.distributed.get_backend() != "nccl")
        acc = torch.zeros(sa.accumulator_len(256), dtype=torch.float64, device=torch.device("cuda", local_rank))
        curve = np.zeros(len(xs))
        lo, hi = D.shard_range(n_rays, rank, world)
        with sa.RayTracer(full, device=local_rank) as rt:
            stream = torch.cuda.Stream(device=acc.device)
            torch.cuda.set_stream(stream)
            rt.set_stream(stream.cuda_stream)
            fixed64 = args.accumulation == "fixed64"
            rt.set_accumulation_mode(args.accumulation)
            for i, x in enumerate(xs):
                D.heartbeat("point %d of %d" % (i + 1, len(xs)))
                if args.mode == "angular":
                    rt.set_telescope_angles(float("nan"), float(x))
                else:
                    rt.set_axion_mass(float(x))
                # accumulate=False: the launch zeroes the accumulator and - fixed64 - fixes the quanta for THIS point (in the gas
                # stage the weight bound follows the mass)
                p = rt.trace_params(hi - lo, ray_id_offset=i * n_rays + lo, flags=flags, accumulate=False)
                rt.trace_histogram_device(p, acc.data_ptr())
                red = acc if use_cuda else acc.cpu()
                D.reduce_accumulator(red, dst=0, fixed64=fixed64)
                if fixed64:      # raw integers -> doubles (every rank: the quanta are the same everywhere; only rank 0's sum is complete)
                    if not use_cuda:
                        acc.copy_(red)
                    rt.finalize_accumulator_device(p, acc.data_ptr())
                    rt.synchronize()   # raises if the integers did not resolve the weights / a slot wrapped
                    red = acc if use_cuda else acc.cpu()
                curve[i] = float(red[256 * 256 + L.ACC["SUM_WEIGHTS"]].item())
                if rank == 0:
                    assert float(red[256 * 256 + L.ACC["N_RAYS"]].item()) == n_rays
            rt.synchronize()
        mine = None
    else:
        mine = D.shard_angles(len(xs), rank, world)
    with sa.RayTracer(full, device=local_rank) if mine is not None else _Null() as rt:
        if mine is None:
            pass
        elif args.mode == "angular" and args.fused:
            # this rank's group of angles through the fused kernel, on the ray ids [0, n_rays) like every other rank's group
            vals = list(sa.performAngularScan(rt, 0, 0, 1, n_rays, flags=flags, angles=xs[mine], fused=True)[1]) if len(mine) else []
        elif args.mode == "angular":
            # every bin keeps its own ray-id block so that the result does not depend on the number of ranks
            vals = [sa.performAngularScan(rt, 0, 0, 1, n_rays, flags=flags, angles=[xs[i]], ray_id_offset=i * n_rays)[1][0] for i in mine]
        else:
            vals = [sa.performAxionMassScanHostLoop(rt, [xs[i]], n_rays, flags=flags, ray_id_offset=i * n_rays)[0] for i in mine]
    if mine is not None:
        dev = torch.device("cuda", local_rank) if (world > 1 and torch.distributed.get_backend() == "nccl") else "cpu"
        curve = D.gather_scan(torch.tensor(vals, dtype=torch.float64, device=dev), mine, len(xs)).cpu().numpy()
    if rank == 0:
        rel = curve / curve.max()
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            if args.mode == "angular":
                ref = tables.reference_curves()
                xmm_x = ref["xmm_angle_arcmin"] / 60.0
                ok = ref["xmm_effective_area"] > 0
                xmm = np.interp(xs, xmm_x[ok], ref["xmm_effective_area"][ok] / ref["xmm_effective_area"].max())
                mcx = np.interp(xs, ref["mcxtrace_angle_deg"], ref["mcxtrace_rel_flux"])
                f.write("Angle [deg],flux,relative flux,XMM theory,McXtrace\n")
                for a, c, r, t, m in zip(xs, curve, rel, xmm, mcx):
                    f.write("%.6g,%.17g,%.8f,%.6f,%.6f\n" % (a, c, r, t, m))
                print("angular scan: relative flux", np.round(rel, 4).tolist())
                print("XMM theory          :", np.round(xmm, 4).tolist())
                print("McXtrace            :", np.round(mcx, 4).tolist())
            else:
                f.write("m_a [eV],flux,relative flux,flux error\n")
                for i, (a, c, r) in enumerate(zip(xs, curve, rel)):
                    f.write("%.8g,%.17g,%.8f,%s\n" % (a, c, r, "%.6g" % errs[i] if errs is not None else ""))
                print("mass scan: m_a", np.round(xs, 5).tolist())
                print("relative flux", np.round(rel, 4).tolist())
        print("wrote", args.out)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    D.report_stage("done")


class _Null:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


if __name__ == "__main__":
    main()
