#!/usr/bin/env python3
"""bench.py — rays/second of the per-ray hot path on N MI355X GPUs of one node.

A "step" is one pass of the hot path (sample -> bore/pipes -> Wolter shells -> reflectivity -> detector ->
focal-plane histogram) over one batch of --rays-per-step rays per GPU, with all tables resident in HBM.
Workload = BASELINE.json configs[2]: BabyIAXO magnet + XMM-Newton shells (58), vacuum, InGridIAXO window,
256x256 focal-plane image; one step = one 1e9-ray image per GPU.  Inputs are the documented synthetic tables
(E1 Primakoff emission on AGSS09, G1 Henke gold reflectivity; the reference's own input files are not shipped).

Multi-GPU: one process per GPU (torchrun contract), rays shard by global ray id, every rank accumulates its own
image and ONE RCCL reduce of the fused accumulator closes the timed region.  --scaling weak (default): every rank
traces --rays-per-step rays per step; --scaling strong: --rays-per-step is the total per step, split over the ranks
by distributed.shard_range (BASELINE configs[4]: "1e10 rays across 8 MI355X" = --scaling strong --rays-per-step 1e9
--steps 10 on 8 ranks).

Roofline block: in cycles the kernel is nearest to the f64 VALU issue wall, not HBM; the clock the full chip is
granted (sclk_mhz, ~2.08 of 2.4 GHz) is what rations the cycles.  PMC counters cannot be read from
inside this process, so the per-ray counter figures come from the committed separate-pass profile of the SAME build
and workload (profiles/pmc_current.json -> profiles/<tag>_*_pmc_summary.json, made by tools/pmc_profile.sh +
tools/pmc_summary.py, which store sart_build_id() of the profiled library) and are combined with the kernel duration
measured live here with HIP events on the launch stream.  If the library being timed has another build id than the
profile, every counter-derived field is null and the note says why.
    achieved  [TFLOP/s]     = f64_flop_per_ray x rays per launch / avg kernel duration      (peak 78.6: f64 vector)
    hbm_achieved_gbs [GB/s] = fabric_bytes_per_ray x rays per launch / avg kernel duration  (peak 8000); hbm_frac = / 8000
    valu_issue_utilisation  = fraction of the SIMDs' cycles in which a VALU instruction issues (from the profile run)

Prints one JSON line on rank 0.  --profile-run skips the CPU baseline and the side workloads (for use under rocprofv3).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
F64_VALU_PEAK_TFLOPS = 78.6    # 256 CU x 4 SIMD x 16 f64 lanes / clk x 2 flop x 2.4 GHz
NOMINAL_SCLK_MHZ = 2400.0
# algorithmic bytes per ray of the REFERENCE's formulation (f64 tables, 11-probe searches, no cache
# credit).  Informational only: the redesigned path (stage A0, guide tables, LDS-resident tables) needs far fewer bytes,
# so this figure does not bound the kernel and is no longer used as `achieved`.
BYTES_KILLED, BYTES_MIRROR, BYTES_DETECTOR = 184.0, 184.0 + 64.0, 456.0

WORKLOADS = {
    "babyiaxo_xmm": "BabyIAXO magnet + XMM-Newton 58 shells, vacuum, InGridIAXO, 256x256 focal-plane image (BASELINE configs[2])",
    "cast_llnl_gold": "CAST magnet + LLNL 14 shells, gold_0.25microns reflectivities (BASELINE configs[1])",
    "cast_llnl": "CAST magnet + LLNL 14 shells, the reference's own pairing: four multilayer coatings by shell group "
                 "(raytracer.nim:1164-1187, computeReflectivity :1571-1580), InGrid2018",
    "babyiaxo_xmm_gas": "BabyIAXO magnet + XMM-Newton shells, gas stage (the m_a-scan kernel variant of BASELINE configs[4])",
    "babyiaxo_xmm_rot": "BabyIAXO magnet + XMM-Newton shells, telescope turned 0.1 deg, effective-area flags, chip 100 mm "
                        "(one angle bin of BASELINE configs[3])",
    "babyiaxo_xmm_small_tables": "diagnostic only: the headline workload on 400 x 300 / 200 x 200 tables that stay in every XCD's L2",
    "babyiaxo_xmm_gas_scan32": "fused m_a scan, 32 masses 0 .. 0.02 eV: BabyIAXO + XMM-Newton shells, gas stage, full AGSS09 emission (all "
                               "terms, tables made on the device) - every ray traced once and weighed for every mass (BASELINE configs[4])",
    "babyiaxo_xmm_ascan16": "fused angular scan, 16 telescope angles 0 .. 0.3 deg: BabyIAXO magnet + XMM-Newton shells, effective-area flags, "
                            "chip 100 mm - every ray sampled and cut once, turned through every angle (BASELINE configs[3])",
}
SCAN_MASSES = 32   # the m_a scan workload: masses linspace(0, 0.02 eV, 32) around m_gamma = 0.008235 eV
SCAN_ANGLES = 16   # the angular scan workload: telescope_turned_y linspace(0, 0.3 deg, 16) = one kernel launch per pass over the rays


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50, help="timed steps (50 x 1e9 rays = 0.9 s of kernel time on one MI355X)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rays-per-step", type=float, default=1e9,
                    help="rays per step: per GPU (--scaling weak) or in total (--scaling strong); 1e9 = one BabyIAXO image of "
                         "BASELINE configs[2]")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--workload", default="babyiaxo_xmm", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-sample", type=float, default=6e8, help="rays of the CPU-baseline sample")
    ap.add_argument("--accumulation", default="f64", choices=["f64", "fixed64"],
                    help="f64 (default): f64 atomics, as the reference adds on the CPU; fixed64: deterministic integer accumulation "
                         "(SART_ACCUM_FIXED64), int64 reduce - image and sums bitwise independent of the number of GPUs")
    ap.add_argument("--headroom", type=int, default=0,
                    help="fixed64: bits of headroom (0 = the library's default, 27: one accumulator then lasts ~1e12 BabyIAXO rays; "
                         "31 carries 2.6e12, profiles/r04_v48_fixed64_long_run.txt)")
    ap.add_argument("--profile-run", action="store_true", help="no CPU baseline / side workloads (run under rocprofv3)")
    ap.add_argument("--no-proof", action="store_true",
                    help="skip the FIXED64 bitwise proof behind the timed region (counter profiles: its 2e8-ray launches of the same "
                         "kernel would be averaged into the profiled ones, tools/pmc_summary.py)")
    ap.add_argument("--emit-bitwise-constants", metavar="PATH", default=None,
                    help="one GPU: write the FIXED64 reference image's SHA-256, integer flux and the hash of the input tables to PATH "
                         "(the committed copy is tests/golden/bench_bitwise_fixed64.json; a --gpus N run holds its reduced image to it)")
    ap.add_argument("--preflight", action="store_true",
                    help="first contact with a multi-GPU node: only bring the process group up (RCCL), reduce one 512 KB buffer "
                         "once and print {world_size, backend, reduce_ms}; no tables, no rays")
    return ap.parse_args()


def make_setup(workload: str):
    """The FullRaytraceSetup and trace flags of a named workload (default full-size tables: 1968 x 1500 emission CDFs,
    1000 x 1000 reflectivity)."""
    import solaraxionraytracing_amd as sa
    from solaraxionraytracing_amd import _lib as L
    flags = 0
    if workload == "babyiaxo_xmm":
        full = sa.initFullSetup()
    elif workload == "babyiaxo_xmm_small_tables":
        full = sa.initFullSetup(n_radii=400, n_energies=300, refl_n_angles=200, refl_n_energies=200)
    elif workload == "cast_llnl_gold":
        full = sa.initFullSetup(L.ES_CAST, L.DK_INGRID2018, L.SK_VACUUM, L.TK_LLNL, reflectivity="gold")
    elif workload == "cast_llnl":
        full = sa.initFullSetup(L.ES_CAST, L.DK_INGRID2018, L.SK_VACUUM, L.TK_LLNL)
    elif workload == "babyiaxo_xmm_gas":
        full = sa.initFullSetup(stage=L.SK_GAS)
    elif workload == "babyiaxo_xmm_gas_scan32":
        full = sa.initFullSetup(stage=L.SK_GAS, emission="agss09-device")
    elif workload in ("babyiaxo_xmm_rot", "babyiaxo_xmm_ascan16"):
        full = sa.initFullSetup()
        full.setup.chip_x_max = full.setup.chip_y_max = 100.0
        if workload == "babyiaxo_xmm_rot":
            full.setup.telescope_turned_y_deg = 0.1
        flags = L.CF_IGNORE_DET_WINDOW | L.CF_IGNORE_GAS_ABS | L.CF_IGNORE_CONV_PROB
    else:
        raise ValueError(workload)
    return full, flags


def active_knobs():
    """SART_* environment variables that change what the library does (libsart reads them in sart_create)."""
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith("SART_")}


def load_pmc(workload: str):
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_current.json")) as f:
            return json.load(f).get(workload)
    except Exception:
        return None


def pmc_for_this_build(workload: str):
    """The committed counter profile of `workload` if it was collected on THE BUILD BEING TIMED (sart_build_id() of the
    loaded libsart.so == the id tools/pmc_summary.py stored with the profile), else (None, why).  A profile of another
    build says nothing about this one: its figures are not reported."""
    from solaraxionraytracing_amd import _lib
    pmc = load_pmc(workload)
    lib_id = _lib.build_id()
    if pmc is None:
        return None, "no committed PMC profile for this workload (profiles/pmc_current.json)", lib_id
    if pmc.get("build_id") != lib_id:
        return None, ("the committed PMC profile %s belongs to build %s, the library being timed is build %s: counter-derived "
                      "fields are null until tools/pmc_profile.sh + tools/pmc_summary.py --publish are re-run on this build"
                      % (pmc.get("source"), pmc.get("build_id"), lib_id)), lib_id
    return pmc, None, lib_id


def side_roofline(*a):
    """roofline_block for an entry of other_workloads: the same figures without the explanatory note (it is the headline's, word for
    word - six copies of it made the line 15 KB long)."""
    blk = roofline_block(*a)
    if blk["achieved"] is not None:      # (without counters of this build the note says why: kept)
        blk["note"] = "as the headline's roofline.note"
    return blk


def roofline_block(workload: str, rays_per_launch: float, avg_kernel_s: float, n_launch: int, summ: dict, total_rays: float):
    """Counter-derived roofline of the trace kernel (see the module docstring for the formulas).  Flat scalars only (the
    driver's parsed record keeps scalars): the f64 issue roofline in achieved / peak / frac, the memory side in hbm_*."""
    pmc, why_not, lib_id = pmc_for_this_build(workload)
    frac_det = summ["N_PASSED_TILL_WINDOW"] / total_rays
    frac_mirror = summ["N_SHELL_SELECTED"] / total_rays - frac_det
    frac_killed = 1.0 - summ["N_SHELL_SELECTED"] / total_rays
    ref_bytes_per_ray = frac_killed * BYTES_KILLED + frac_mirror * BYTES_MIRROR + frac_det * BYTES_DETECTOR
    blk = {"bound": "f64-valu-issue", "achieved": None, "peak": F64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": None,
           "traffic": None, "kernel": "trace_angular_scan_kernel" if workload == "babyiaxo_xmm_ascan16" else "trace_histogram_kernel", "avg_kernel_ms": avg_kernel_s * 1e3, "launches": n_launch,
           "rays_per_launch": rays_per_launch,
           "hbm_frac": None, "hbm_achieved_gbs": None, "hbm_peak_gbs": HBM_PEAK_GBS, "fabric_bytes_per_ray": None,
           "achieved_active_lanes": None, "frac_active_lanes": None,
           "valu_issue_utilisation": None, "valu_lane_utilisation": None, "valu_insts_per_64_rays": None, "f64_flop_per_ray": None,
           "l2_hit_rate": None, "build_id": lib_id, "pmc_build_id": None, "pmc_source": None,
           "reference_formulation_bytes_per_ray": ref_bytes_per_ray,
           # SURVEY 8(d)'s own formula, evaluated as written: above 1 by construction of this design (see the note) - reported so
           # that nobody has to recompute it, never as `frac`
           "reference_formulation_frac_of_hbm_peak": ref_bytes_per_ray * rays_per_launch / avg_kernel_s / 1e9 / HBM_PEAK_GBS,
           "note": "bound: in cycles the kernel is nearest to the vector-issue wall (valu_issue_utilisation of the SIMDs' cycles "
                   "issue a vector instruction); the cycles themselves are rationed: with all 256 CUs in this kernel the power "
                   "management grants sclk_mhz of 2400 (128 CUs of the same kernel run at 2370, profiles/*_power_cu_mask.txt), so what "
                   "moves the rate is energy per ray (nj_per_ray), DESIGN.md 4.  frac = achieved / peak of the f64 vector pipe at 2.4 "
                   "GHz is low by construction (42 % of the vector instructions are f64 arithmetic; frac_at_granted_clock: against "
                   "the peak at sclk_mhz) and hbm_frac is the L2-miss traffic of random 4-32-byte gathers fetched as 128-byte "
                   "lines - neither is a wall by itself.  achieved = f64 flop per ray issued (PMC: (2 FMA + "
                   "MUL + ADD + TRANS) x 64 lanes / rays) x rays per launch / live HIP-event kernel duration; *_active_lanes = the "
                   "same x valu_lane_utilisation (lanes switched off by EXEC do no work); traffic = fabric bytes per launch "
                   "(TCC_EA0 read requests by size + WRITE_SIZE, Infinity-Cache hits included: upper bound on HBM bytes); "
                   "hbm_achieved_gbs = traffic / kernel duration.  Counter figures come from the committed separate-pass profile "
                   "of the same build (pmc_build_id == build_id).  reference_formulation_bytes_per_ray is SURVEY 8(d)'s probe "
                   "count of the reference's formulation; the redesigned path does not move those bytes: informational only."}
    if pmc is None:
        blk["note"] = why_not + ".  " + blk["note"]
        return blk
    tflops = pmc["f64_flop_per_ray"] * rays_per_launch / avg_kernel_s / 1e12
    blk["achieved"] = tflops
    blk["frac"] = tflops / F64_VALU_PEAK_TFLOPS
    lane_util = pmc.get("valu_lane_utilisation")
    if lane_util is not None:
        blk["achieved_active_lanes"] = tflops * lane_util
        blk["frac_active_lanes"] = tflops * lane_util / F64_VALU_PEAK_TFLOPS
    blk["valu_issue_utilisation"] = pmc.get("valu_issue_utilisation")
    blk["valu_lane_utilisation"] = lane_util
    blk["valu_insts_per_64_rays"] = pmc.get("valu_insts_per_64_rays")
    blk["f64_flop_per_ray"] = pmc["f64_flop_per_ray"]
    blk["l2_hit_rate"] = pmc.get("l2_hit_rate")
    blk["pmc_source"] = pmc.get("source")
    blk["pmc_build_id"] = pmc.get("build_id")
    if pmc.get("fabric_bytes_per_ray") is not None:
        traffic = pmc["fabric_bytes_per_ray"] * rays_per_launch
        gbs = traffic / avg_kernel_s / 1e9
        blk["traffic"] = traffic
        blk["fabric_bytes_per_ray"] = pmc["fabric_bytes_per_ray"]
        blk["hbm_achieved_gbs"] = gbs
        blk["hbm_frac"] = gbs / HBM_PEAK_GBS
        assert blk["hbm_frac"] <= 1.0, blk
    assert blk["frac"] <= 1.0, blk
    assert blk["valu_issue_utilisation"] is None or blk["valu_issue_utilisation"] <= 1.0, blk
    return blk


def check_image(host, n_img: int, summ: dict):
    """The image must hold the weights the scalars say were accumulated: guards against a timed region with the
    accumulation skipped."""
    img_sum = float(host[:n_img].sum())
    if summ["N_OUTSIDE_IMAGE"] == 0:
        assert abs(img_sum - summ["SUM_WEIGHTS"]) <= 1e-9 * abs(summ["SUM_WEIGHTS"]), (img_sum, summ["SUM_WEIGHTS"])
    else:
        assert 0.0 < img_sum <= summ["SUM_WEIGHTS"] * (1.0 + 1e-9), (img_sum, summ["SUM_WEIGHTS"])
    assert summ["N_PASSED"] > 0 and img_sum > 0.0
    return img_sum


BITWISE_RAYS = 200_000_000     # the id range [0, 2e8) of the bitwise proof (seed 299792458): 2.3 ms of kernel time on one GPU
BITWISE_CONSTANTS = os.path.join(ROOT, "tests", "golden", "bench_bitwise_fixed64.json")


def tables_sha256(full) -> str:
    """Identity of everything the image is a function of on the input side: the flattened setup and every table handed to the
    library.  (The synthetic tables come out of numpy / libm on the host: a host whose libm rounds one entry differently makes
    other inputs - then the committed constants do not apply, and the line says so instead of reporting a mismatch.)"""
    import hashlib
    import numpy as np
    h = hashlib.sha256()
    h.update(bytes(full.setup))
    for a in (full.energies, full.fluxRadiusCDF, full.diffFluxCDFs, full.reflectivity.data, full.detector_tables.x_kev,
              full.detector_tables.strongback, full.detector_tables.window, full.detector_tables.gas_x_kev,
              full.detector_tables.gas_absorption):
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def bitwise_proof(rt, full, flags, rank: int, world: int, dev, seed: int, D, restore=("f64", 0)):
    """The multi-GPU leg proving its RESULT: rays are independent (raytracer.nim:2234) and in
    SART_ACCUM_FIXED64 every sum is an integer, so the image of a fixed ray-id range must not depend on how many ranks traced
    it - to the last bit.  Every rank traces its share of the ids [0, BITWISE_RAYS) (distributed.shard_range: the strong
    sharding), ONE int64 reduce to rank 0, then rank 0
      (a) traces the whole range alone on its own GPU and compares the two raw accumulators slot for slot
          after sart_finalize_accumulator_device (image, flux, sums, counters as doubles)
          -> multi_rank_bitwise_equal_to_single_gpu (world 1: the range as three launches against one launch);
      (b) hashes the image and compares SHA-256 + the flux's bit pattern + N_PASSED with the constants committed
          from a one-GPU run (tests/golden/bench_bitwise_fixed64.json) -> matches_committed_constants (null when this
          host built other input tables than the host the constants come from).
    Outside the timed region.  Returns the block (rank 0) or None."""
    import hashlib
    import numpy as np
    import torch
    import solaraxionraytracing_amd as sa
    from solaraxionraytracing_amd import _lib as L
    rt.set_accumulation_mode("fixed64")
    try:
        n_acc = sa.accumulator_len(256)
        acc = torch.zeros(n_acc, dtype=torch.float64, device=dev)
        lo, hi = D.shard_range(BITWISE_RAYS, rank, world)
        if world == 1:
            cuts = (0, 1, BITWISE_RAYS // 3, BITWISE_RAYS)
            for a, b in zip(cuts[:-1], cuts[1:]):
                rt.trace_histogram_device(rt.trace_params(b - a, seed=seed, ray_id_offset=a, accumulate=True, flags=flags), acc.data_ptr())
        else:
            rt.trace_histogram_device(rt.trace_params(hi - lo, seed=seed, ray_id_offset=lo, accumulate=True, flags=flags), acc.data_ptr())
        D.reduce_accumulator(acc, dst=0, fixed64=True)
        if rank != 0:
            return None
        alone = torch.zeros(n_acc, dtype=torch.float64, device=dev)
        rt.trace_histogram_device(rt.trace_params(BITWISE_RAYS, seed=seed, accumulate=True, flags=flags), alone.data_ptr())
        # both through the finalize kernel (integers -> doubles, exact products with the power-of-two quanta), then bit for bit.
        # (The RAW arrays may differ in representation: the six two-limb sums are (hi 2^40 + lo) with lo < 2^40 after a launch,
        # and a reduce adds the ranks' limbs without carrying - the same integer, other limbs.)
        rt.finalize_accumulator_device(rt.trace_params(1), acc.data_ptr())
        rt.finalize_accumulator_device(rt.trace_params(1), alone.data_ptr())
        rt.synchronize()
        n_diff = int((acc.view(torch.int64) != alone.view(torch.int64)).sum().item())
        host = acc.cpu().numpy()
    finally:
        rt.set_accumulation_mode(*restore)
    n_img = 256 * 256
    summ = {k: float(host[n_img + i]) for k, i in L.ACC.items()}
    blk = {"rays": BITWISE_RAYS, "seed": seed, "ranks": world, "slots_that_differ": n_diff, "equal": n_diff == 0,
           "image_sha256": hashlib.sha256(host[:n_img].tobytes()).hexdigest(), "sum_weights_hex": float(summ["SUM_WEIGHTS"]).hex(),
           "sum_weights": summ["SUM_WEIGHTS"], "n_passed": int(summ["N_PASSED"]), "n_rays": int(summ["N_RAYS"]),
           "tables_sha256": tables_sha256(full), "matches_committed_constants": None, "committed": None}
    assert blk["n_rays"] == BITWISE_RAYS, blk
    try:
        with open(BITWISE_CONSTANTS) as f:
            want = json.load(f)
    except Exception:
        want = None
    if want is not None:
        if want.get("tables_sha256") != blk["tables_sha256"]:
            blk["committed"] = "not applicable: this host built other input tables (tables_sha256) than the host the constants come from"
        else:
            blk["committed"] = os.path.relpath(BITWISE_CONSTANTS, ROOT)
            blk["matches_committed_constants"] = all(want[k] == blk[k] for k in ("image_sha256", "sum_weights_hex", "n_passed", "rays", "seed"))
    return blk


def flat_scalars(out: dict) -> dict:
    """BASELINE's whole metric as top-level scalars (the driver's parsed record keeps scalars and dicts of scalars; the lists
    and nested blocks of this line do not survive it): the second metric - the effective-area curve's RMS against the CPU
    reference, raytracer.nim:2799-2802 - and the rate of every other configuration of BASELINE.json."""
    flat = {}
    ea = out.get("effective_area_rms")
    if ea:
        flat["effective_area_rms_value"] = ea["value"]
        flat["effective_area_rms_fused_value"] = ea["fused_scan_value"]
    for w in out.get("other_workloads", []):
        name = w["workload"]
        if name == WORKLOADS["cast_llnl_gold"]:
            flat["cast_llnl_gold_rays_per_s"] = w["rays_per_s"]
        elif name == WORKLOADS["cast_llnl"]:
            flat["cast_llnl_4coatings_rays_per_s"] = w["rays_per_s"]
        elif name == WORKLOADS["babyiaxo_xmm_gas"]:
            flat["gas_rays_per_s"] = w["rays_per_s"]
        elif name == WORKLOADS["babyiaxo_xmm_rot"]:
            flat["rot_rays_per_s"] = w["rays_per_s"]
        elif name == WORKLOADS["babyiaxo_xmm_gas_scan32"]:
            flat["scan32_ray_mass_per_s"] = w["ray_mass_evaluations_per_s"]
            flat["scan32_speedup_over_host_loop"] = w["speedup_over_host_loop"]
        elif name == WORKLOADS["babyiaxo_xmm_ascan16"]:
            flat["ascan16_ray_angle_per_s"] = w["ray_angle_evaluations_per_s"]
            flat["ascan16_speedup_over_host_loop"] = w["speedup_over_host_loop"]
            m = w["sharding_model"]
            flat["ascan_t_shared_ps_per_ray"] = m["t_shared_ps_per_ray"]
            flat["ascan_t_angle_ps_per_ray_angle"] = m["t_angle_ps_per_ray_angle"]
            flat["ascan50_by_ray_id_g8_ps_per_ray_predicted"] = m["by_ray_id_g8_ps_per_ray"]
            flat["ascan50_by_angle_g8_ps_per_ray_predicted"] = m["by_angle_g8_ps_per_ray"]
        elif "cells_per_s" in w:
            flat["emission_table_cells_per_s"] = w["cells_per_s"]
    rec = out.get("record_interface")
    if rec:
        flat["records_every_rays_per_s"] = rec["every_record"]["rays_per_s"]
        flat["records_passed_rays_per_s"] = rec["passed_records_only"]["rays_per_s"]
    det = out.get("deterministic_accumulation")
    if det:
        flat["fixed64_vs_f64_rate"] = det["vs_f64_rate"]
        flat["fixed64_bitwise_equal_across_launch_split_and_replicas"] = det["bitwise_equal_across_launch_split_and_replicas"]
    cpu = out.get("cpu_baseline")
    if cpu:
        flat["cpu_rays_per_s"] = cpu["value"]
        flat["cpu_cores"] = cpu["cores"]
        flat["cpu_1t_rays_per_s"] = cpu.get("single_thread_rays_per_s")
        flat["gpu_over_cpu"] = out["value"] / cpu["value"]
    bw = out.get("bitwise_proof")
    if bw:
        flat["multi_rank_bitwise_equal_to_single_gpu"] = bw["equal"] if out["n_gpus"] > 1 else None
        flat["fixed64_launch_split_bitwise_equal"] = bw["equal"] if out["n_gpus"] == 1 else None
        flat["fixed64_image_matches_committed_constants"] = bw["matches_committed_constants"]
    return flat


def main():
    args = parse()
    from solaraxionraytracing_amd import distributed as D

    # `--gpus N` means N ranks whoever starts this script: under torchrun (WORLD_SIZE set) this process is one of them and
    # WORLD_SIZE must equal N; stand-alone with N > 1 this process only starts N copies of itself (one rank each) and
    # waits for them.  Either way a mismatch, or fewer devices than ranks, ends with exit code 2 BEFORE anything touches a
    # GPU - a line whose n_gpus is not what was asked for is never printed.
    # SART_BENCH_BACKEND=gloo + SART_BENCH_DEVICE=0: rehearsal of the multi-rank path on a one-GPU box.
    gloo_preflight = args.preflight and os.environ.get("SART_BENCH_BACKEND") == "gloo"   # touches no card: runs anywhere
    rc = D.launch_ranks_if_needed(args.gpus, os.path.abspath(__file__), sys.argv[1:], need_devices=not gloo_preflight)
    if rc is not None:
        raise SystemExit(rc)
    if args.preflight:
        res = D.preflight(os.environ.get("SART_BENCH_BACKEND"))
        assert res["world_size"] == args.gpus, (res, args.gpus)
        if int(os.environ.get("RANK", "0")) == 0:
            print(json.dumps(res))
        raise SystemExit(0 if res["preflight"] == "ok" else 4)

    import torch
    import torch.distributed as dist

    import solaraxionraytracing_amd as sa
    from solaraxionraytracing_amd import _lib as L

    rank, world, local_rank = D.init_process_group_from_env(os.environ.get("SART_BENCH_BACKEND"))
    assert world == args.gpus, (world, args.gpus)
    backend = dist.get_backend() if world > 1 else "none"
    if "SART_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["SART_BENCH_DEVICE"])
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    full, flags = make_setup(args.workload)
    wl_name = WORKLOADS[args.workload]
    rays_rank, step_total, lo_in_step = D.step_shard(args.scaling, int(args.rays_per_step), rank, world)
    rt = sa.RayTracer(full, device=local_rank)
    # Launches, torch ops on the accumulator and the RCCL reduce are ordered by ONE explicit stream.  (torch's
    # default stream has handle 0, which sart_set_stream reads as "the context's own stream".)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    rt.set_stream(stream.cuda_stream)
    fixed64 = args.accumulation == "fixed64"
    rt.set_accumulation_mode(args.accumulation, args.headroom)
    import numpy as np
    scan_masses = np.linspace(0.0, 0.02, SCAN_MASSES) if args.workload == "babyiaxo_xmm_gas_scan32" else None
    scan_angles = np.linspace(0.0, 0.3, SCAN_ANGLES) if args.workload == "babyiaxo_xmm_ascan16" else None
    # 8-byte slots: doubles, or int64 in fixed64 mode.  Image accumulator - or, for the scans, (points + 1) rows of 8 slots
    acc_len = sa.mass_scan_len(SCAN_MASSES) if scan_masses is not None else sa.angular_scan_len(SCAN_ANGLES) if scan_angles is not None else sa.accumulator_len(256)
    acc = torch.zeros(acc_len, dtype=torch.float64, device=dev)
    # fixed64, image workloads: a second limb per slot (sart_rollover_accumulator_device), folded every 64 steps - a run of any
    # length at the default headroom (2500 steps = 2.5e12 rays used to need --headroom 31)
    roll = fixed64 and scan_masses is None and scan_angles is None
    hi_limbs = torch.zeros(acc_len, dtype=torch.int64, device=dev) if roll else None
    seed = 299792458

    def step(k: int):
        # global ray ids: step-major, rank-minor => the union over ranks and steps is a contiguous id range
        offset = k * step_total + lo_in_step
        p = rt.trace_params(rays_rank, seed=seed, ray_id_offset=offset, accumulate=True, flags=flags)
        if scan_masses is not None:
            rt.trace_mass_scan_device(p, scan_masses, acc.data_ptr())   # one pass over the rays, every mass
        elif scan_angles is not None:
            rt.trace_angular_scan_device(p, scan_angles, acc.data_ptr())   # one pass over the rays, every angle
        else:
            rt.trace_histogram_device(p, acc.data_ptr())
            if roll and (k + 1) % 64 == 0:
                rt.rollover_accumulator_device(p, acc.data_ptr(), hi_limbs.data_ptr())

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # Progress reports for the self-launcher's stall clock (distributed.heartbeat: every live rank silent for SART_STALL_TIMEOUT ->
    # exit 3 with the last report of each rank; a file write, no-op without a launcher - outside the device's work either way)
    D.heartbeat("warmup")
    for k in range(args.warmup):
        step(10_000 + k)   # ray ids outside the timed range
    if world > 1:  # warm the communicator
        D.reduce_accumulator(acc.clone(), dst=0, fixed64=fixed64)
    barrier()
    acc.zero_()
    if roll:
        hi_limbs.zero_()
    rt.enable_kernel_timing(True)
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
        if (k & 63) == 63:
            D.heartbeat("step %d of %d queued" % (k + 1, args.steps))
    stream.synchronize()                 # this rank's launches are done (the reduce would wait for them anyway)
    t_red = time.perf_counter()
    D.heartbeat("reduce")
    if roll:                             # fold once more: every slot < 2^40 on every rank, then both limb arrays reduce as int64 sums
        rt.rollover_accumulator_device(rt.trace_params(1), acc.data_ptr(), hi_limbs.data_ptr())
        D.reduce_accumulator(hi_limbs, dst=0, fixed64=True)
    D.reduce_accumulator(acc, dst=0, fixed64=fixed64)     # the single RCCL reduce of the output histograms
    if fixed64 and rank == 0:            # raw integer accumulator -> doubles, in place (part of the timed region)
        if roll:
            rt.finalize_accumulator_limbs_device(rt.trace_params(1), acc.data_ptr(), hi_limbs.data_ptr())
        elif scan_masses is not None:
            rt.finalize_mass_scan_device(rt.trace_params(1, flags=flags), scan_masses, acc.data_ptr())
        elif scan_angles is not None:
            rt.finalize_angular_scan_device(rt.trace_params(1, flags=flags), SCAN_ANGLES, acc.data_ptr())
        else:
            rt.finalize_accumulator_device(rt.trace_params(1), acc.data_ptr())
    barrier()
    t1 = time.perf_counter()
    D.heartbeat("timed region done")
    rt.synchronize()                     # raises what the FIXED64 finalize found (unresolved weights, a wrapped slot)
    # Only rank 0 finalizes, so only rank 0 can raise here - the other ranks walk into the all_reduce below and wait for it.  Under
    # the self-launcher that ends well (it sees rank 0's exit code, names the rank and ends the others: distributed.py).  The knob
    # rehearses exactly that on the gloo backend (tests/test_distributed_cpu.py, tests/test_gpu_parity.py); never honoured on RCCL.
    if rank == 0 and os.environ.get("SART_BENCH_FAIL_RANK0_AFTER_REDUCE") and os.environ.get("SART_BENCH_BACKEND") == "gloo":
        raise RuntimeError("SART_BENCH_FAIL_RANK0_AFTER_REDUCE: rank 0 fails behind the reduce (rehearsal)")
    # MAX over ranks of the whole timed region; of the reduce (its wait for the slowest rank included) rank 0's figure
    times = torch.tensor([t1 - t0, t_red - t0], dtype=torch.float64, device=dev)
    rays_all = torch.zeros(world, dtype=torch.float64, device=dev)
    rays_all[rank] = float(rays_rank)
    if world > 1:
        if backend != "nccl":
            times, rays_all = times.cpu(), rays_all.cpu()
        dist.all_reduce(times, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays_all, op=dist.ReduceOp.SUM)
    elapsed_s = float(times[0].item())
    reduce_ms = (t1 - t_red) * 1e3
    kernel_ms, n_launch = rt.kernel_timing()
    rt.enable_kernel_timing(False)
    # the result, not only the rate: a fixed id range traced by all ranks in FIXED64 against rank 0 alone (every rank takes part)
    D.heartbeat("bitwise proof")
    proof = None
    if scan_masses is None and scan_angles is None and full.fluxRadiusCDF is not None and not args.no_proof:
        proof = bitwise_proof(rt, full, flags, rank, world, dev, seed, D, restore=(args.accumulation, args.headroom))
    proof_failed = False

    if rank == 0:
        host = acc.cpu().numpy()
        n_img = 256 * 256
        total_rays = float(args.steps) * step_total
        scan_block = None
        ascan_block = None
        if scan_angles is not None:
            # the scan's own checks: every ray counted once, every angle got weights, the curve falls off with the angle
            per_angle, shared = sa.split_angular_scan(host, SCAN_ANGLES)
            assert shared["N_RAYS"] == total_rays, (shared, total_rays)
            assert per_angle["N_PASSED"].min() > 0 and per_angle["SUM_WEIGHTS"].min() > 0.0
            assert int(np.argmax(per_angle["SUM_WEIGHTS"])) <= 1 and per_angle["SUM_WEIGHTS"][-1] < per_angle["SUM_WEIGHTS"][0]
            summ = {"SUM_WEIGHTS": float(per_angle["SUM_WEIGHTS"][0]), "N_PASSED": float(per_angle["N_PASSED"][0]),
                    "N_PASSED_TILL_WINDOW": float(per_angle["N_PASSED_TILL_WINDOW"].mean()), "N_SHELL_SELECTED": float(per_angle["N_SHELL_SELECTED"].mean()),
                    "N_REACHED_TELESCOPE": shared["N_REACHED_TELESCOPE"]}
            img_sum = summ["SUM_WEIGHTS"]
            ascan_block = {"angles": SCAN_ANGLES, "turned_y_deg": [round(float(a), 6) for a in scan_angles],
                           "ray_angle_evaluations_per_s": total_rays * SCAN_ANGLES / elapsed_s,
                           "relative_flux": [round(float(x), 6) for x in per_angle["SUM_WEIGHTS"] / per_angle["SUM_WEIGHTS"].max()],
                           "relative_error_on_axis": float(np.sqrt(per_angle["SUM_WEIGHTS_SQ"][0]) / per_angle["SUM_WEIGHTS"][0])}
        elif scan_masses is None:
            summ = {k: float(host[n_img + i]) for k, i in L.ACC.items()}
            assert summ["N_RAYS"] == total_rays, (summ["N_RAYS"], total_rays)
            img_sum = check_image(host, n_img, summ)
        else:
            # the scan's own checks: every ray counted once, every mass got weights, the curve peaks at the resonance
            per_mass, shared = sa.split_mass_scan(host, SCAN_MASSES)
            assert shared["N_RAYS"] == total_rays, (shared, total_rays)
            assert per_mass["N_PASSED"].min() > 0 and per_mass["SUM_WEIGHTS"].min() > 0.0
            k_res = int(np.argmin(np.abs(scan_masses - 0.008235)))
            assert int(np.argmax(per_mass["SUM_WEIGHTS"])) == k_res, per_mass["SUM_WEIGHTS"].tolist()
            k0 = int(np.argmax(per_mass["SUM_WEIGHTS"]))
            summ = {"SUM_WEIGHTS": float(per_mass["SUM_WEIGHTS"][k0]), "N_PASSED": float(per_mass["N_PASSED"][k0]),
                    "N_PASSED_TILL_WINDOW": float(shared["N_ON_DETECTOR"]), "N_SHELL_SELECTED": shared["N_SHELL_SELECTED"],
                    "N_REACHED_TELESCOPE": shared["N_REACHED_TELESCOPE"]}
            img_sum = summ["SUM_WEIGHTS"]
            scan_block = {"masses": SCAN_MASSES, "m_a_ev": [round(float(m), 6) for m in scan_masses],
                          "ray_mass_evaluations_per_s": total_rays * SCAN_MASSES / elapsed_s,
                          "relative_flux": [round(float(x), 6) for x in per_mass["SUM_WEIGHTS"] / per_mass["SUM_WEIGHTS"].max()],
                          "relative_error_at_resonance": float(np.sqrt(per_mass["SUM_WEIGHTS_SQ"][k0]) / per_mass["SUM_WEIGHTS"][k0])}
        value = total_rays / elapsed_s
        avg_kernel_s = kernel_ms / 1e3 / max(1, n_launch)
        out = {
            "metric": "rays/sec",
            "value": value,
            "unit": "rays/s",
            "n_gpus": world,
            "world_size": world,
            "backend": backend,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed_s / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "accumulation": args.accumulation,
            "data": "synthetic",
            "reduce_ms": reduce_ms,
            "slowest_rank_compute_ms_per_step": float(times[1].item()) / args.steps * 1e3,
            "config": {"workload": wl_name, "rays_per_step_per_gpu": rays_rank,
                       "rays_per_step_per_rank": [int(x) for x in rays_all.tolist()], "total_rays": total_rays,
                       "tables": full.meta, "sharding": "global ray id, 1 RCCL reduce of image+scalars",
                       "device": rt.device_info(), "knobs": active_knobs()},
            "roofline": roofline_block(args.workload, float(rays_rank), avg_kernel_s, n_launch, summ, total_rays),
            "results": {"flux": summ["SUM_WEIGHTS"], "image_sum": img_sum, "passed_fraction": summ["N_PASSED"] / total_rays,
                        "reached_telescope_fraction": summ["N_REACHED_TELESCOPE"] / total_rays,
                        "shell_selected_fraction": summ["N_SHELL_SELECTED"] / total_rays},
        }
        if proof is not None:
            out["bitwise_proof"] = proof
            # (a run that WRITES the constants is not held to the ones it replaces)
            proof_failed = not proof["equal"] or (proof["matches_committed_constants"] is False and not args.emit_bitwise_constants)
            if args.emit_bitwise_constants and world == 1:
                with open(args.emit_bitwise_constants, "w") as f:
                    json.dump({k: proof[k] for k in ("rays", "seed", "image_sha256", "sum_weights_hex", "sum_weights", "n_passed",
                                                     "tables_sha256")} | {"workload": wl_name, "build_id": L.build_id(),
                               "made_by": "python bench.py --emit-bitwise-constants PATH (one MI355X)"}, f, indent=1)
                    f.write("\n")
        if scan_block is not None:
            out["mass_scan"] = scan_block
        if ascan_block is not None:
            out["angular_scan"] = ascan_block
        if world == 1 and not args.profile_run:
            en = energy_block(step, stream, float(rays_rank))
            out["roofline"].update({"socket_power_w": en["socket_power_w"], "sclk_mhz": en["sclk_mhz"], "nj_per_ray": en["nj_per_ray"]})
            if en["sclk_mhz"] and out["roofline"]["achieved"] is not None:
                # the peak is a 2.4 GHz figure; with all 256 CUs in this kernel the power management grants ~2.07 GHz
                peak_granted = F64_VALU_PEAK_TFLOPS * en["sclk_mhz"] / NOMINAL_SCLK_MHZ
                out["roofline"].update({"peak_at_granted_clock": peak_granted, "frac_at_granted_clock": out["roofline"]["achieved"] / peak_granted})
            out["energy"] = en
            out["cpu_baseline"] = cpu_baseline(full, int(args.cpu_sample), seed)
            if args.workload == "babyiaxo_xmm":
                out["other_workloads"] = [other_workload_rate(w) for w in ("cast_llnl_gold", "cast_llnl", "babyiaxo_xmm_gas", "babyiaxo_xmm_rot")]
                out["other_workloads"].append(mass_scan_rate())
                out["other_workloads"].append(angular_scan_rate())
                out["other_workloads"].append(emission_table_rate())
                out["deterministic_accumulation"] = fixed64_block(full, value)
                out["record_interface"] = record_interface_block(full)
                out["effective_area_rms"] = effective_area_rms()
        # twice: at the top level, and inside `roofline` - of BENCH_r05.json's `parsed` record only the contract's keys kept their
        # values (top-level extras such as reduce_ms survived as NAMES under extra_keys), while every scalar inside the
        # `roofline`, `config` and `cpu_baseline` dicts did
        flat = flat_scalars(out)
        out.update(flat)
        out["roofline"].update({"also_" + k: v for k, v in flat.items()})
        print(json.dumps(out))
    rt.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    D.report_stage("done")
    if proof_failed:
        # the line above says which comparison failed (bitwise_proof); a run whose reduced image is not the single-GPU image is
        # not a measurement of this path: non-zero exit (the self-launcher hands rank 0's code on)
        sys.stderr.write("bench.py: bitwise proof FAILED: %s\n" % json.dumps(proof))
        raise SystemExit(5)


def energy_block(step, stream, rays_per_step: float, seconds: float = 5.0):
    """What the counters do not show: the clock the power manager grants this kernel (~2.08 of 2.4 GHz at ~1140 W, DESIGN.md
    4).  Outside the timed region the same steps run for `seconds` while rocm-smi is read a few times from a side
    thread: socket power, shader clock, and the energy per ray they imply at the rate of that window.  Best effort: null
    fields if rocm-smi is not there or may not be read."""
    import re
    import subprocess
    import threading
    import torch
    samples = []

    def sampler():
        time.sleep(1.5)
        for _ in range(4):
            try:
                txt = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
                pw = re.search(r"Package Power \(W\): ([0-9.]+)", txt)
                ck = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", txt)
                if pw and ck:
                    samples.append((float(pw.group(1)), float(ck.group(1))))
            except Exception:
                pass
            time.sleep(0.5)

    th = threading.Thread(target=sampler)
    th.start()
    stream.synchronize()
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(8):
            step(20_000 + k)
            k += 1
        stream.synchronize()
    dt = time.perf_counter() - t0
    th.join()
    rate = k * rays_per_step / dt
    blk = {"socket_power_w": None, "sclk_mhz": None, "nj_per_ray": None, "rays_per_s_sustained": rate, "seconds": dt,
           "note": "rocm-smi beside %d untimed steps; for scale: the socket draws 336 W with all waves asleep and 970 W with "
                   "v_fma_f64 alone on every SIMD at 2.4 GHz (tools/microbench/energy_rates.hip, DESIGN.md 4)" % k}
    if samples:
        pw = sum(s[0] for s in samples) / len(samples)
        blk.update({"socket_power_w": pw, "sclk_mhz": sum(s[1] for s in samples) / len(samples), "nj_per_ray": pw / rate * 1e9,
                    "samples": len(samples)})
    return blk


def other_workload_rate(workload: str, n: int = 100_000_000, launches: int = 20):
    """The other kernel variants / BASELINE configs beside the headline workload: `launches` 1e8-ray launches each (BASELINE
    configs[1] is quoted at 1e8 rays) behind five untimed ones - the first launches after an idle gap run while the clock
    is still ramping up -, HIP-event kernel time, own roofline block from that workload's committed PMC profile.
    Informational; `value` is the BabyIAXO workload."""
    import solaraxionraytracing_amd as sa
    from solaraxionraytracing_amd import _lib as L
    full, flags = make_setup(workload)
    with sa.RayTracer(full) as rt:
        for k in range(5):
            rt.trace_histogram(n, seed=2, ray_id_offset=k * n, flags=flags)
        rt.enable_kernel_timing(True)
        for k in range(launches):
            img, s = rt.trace_histogram(n, seed=1, ray_id_offset=k * n, accumulate=(k > 0), flags=flags)
        ms, n_launch = rt.kernel_timing()
    total = float(n) * launches
    assert s["N_RAYS"] == total, (s["N_RAYS"], total)
    import numpy as np
    host = np.concatenate([img.ravel(), np.array([s[k] for k in sorted(L.ACC, key=L.ACC.get)])])
    check_image(host, img.size, s)
    avg_s = ms / 1e3 / n_launch
    return {"workload": WORKLOADS[workload], "rays_per_s": n / avg_s, "ms_per_launch": ms / n_launch,
            "passed_fraction": s["N_PASSED"] / s["N_RAYS"], "roofline": side_roofline(workload, float(n), avg_s, n_launch, s, total)}


def mass_scan_rate(n: int = 1_000_000_000, host_loop_points: int = 4):
    """BASELINE configs[4] through the fused scan kernel: 32 masses on 1e9 rays of the full AGSS09 workload, against the
    reference-shaped host loop (set the mass, re-trace, sum: timed on `host_loop_points` of the masses, same ray count each).
    Unit: (ray, mass) evaluations per second."""
    import numpy as np
    import solaraxionraytracing_amd as sa
    full, flags = make_setup("babyiaxo_xmm_gas_scan32")
    masses = np.linspace(0.0, 0.02, SCAN_MASSES)
    with sa.RayTracer(full) as rt:
        rt.trace_mass_scan(masses, 50_000_000, seed=2)            # clocks up, tables in cache
        rt.enable_kernel_timing(True)
        t0 = time.perf_counter()
        per_mass, shared = rt.trace_mass_scan(masses, n, seed=1)
        wall = time.perf_counter() - t0
        ms, n_launch = rt.kernel_timing()
        pick = np.linspace(0, SCAN_MASSES - 1, host_loop_points).astype(int)
        rt.trace_histogram(50_000_000, seed=2)
        rt.kernel_timing()                                         # (resets: the warm-up launch is not timed)
        loop = []
        for k in pick:
            rt.set_axion_mass(float(masses[k]))
            _, s = rt.trace_histogram(n, seed=1)
            loop.append(s["SUM_WEIGHTS"])
        ms_loop, n_loop = rt.kernel_timing()
        rt.enable_kernel_timing(False)
    assert shared["N_RAYS"] == n and n_launch == (SCAN_MASSES + 31) // 32 and n_loop == host_loop_points
    rel = np.abs(per_mass["SUM_WEIGHTS"][pick] / np.array(loop) - 1.0).max()
    assert rel < 1e-12, rel                                        # the same rays, the same weights
    fused = n * SCAN_MASSES / (ms / 1e3)
    host_loop = n / (ms_loop / 1e3 / n_loop)                       # (ray, mass) evaluations per second of one re-trace per mass
    avg_s = ms / 1e3 / n_launch
    summ = {"N_PASSED_TILL_WINDOW": shared["N_ON_DETECTOR"], "N_SHELL_SELECTED": shared["N_SHELL_SELECTED"]}
    return {"workload": WORKLOADS["babyiaxo_xmm_gas_scan32"], "ray_mass_evaluations_per_s": fused, "rays_per_s": n / (ms / 1e3),
            "ms_per_scan": ms, "launches_per_scan": n_launch, "wall_ms_per_scan": wall * 1e3,
            "host_loop_ray_mass_evaluations_per_s": host_loop, "host_loop_ms_per_mass": ms_loop / n_loop,
            "speedup_over_host_loop": fused / host_loop, "max_rel_diff_to_host_loop": float(rel),
            "passed_fraction": float(per_mass["N_PASSED"].max() / n),
            "roofline": side_roofline("babyiaxo_xmm_gas_scan32", float(n), avg_s, n_launch, summ, float(n))}


def angular_scan_rate(n: int = 200_000_000, host_loop_points: int = 4):
    """BASELINE configs[3] through the fused scan kernel: 16 telescope angles on `n` rays, against the reference-shaped host loop
    (set the angle, re-trace with a flux-only launch, sum: timed on `host_loop_points` of the angles, same ray ids each).
    Unit: (ray, angle) evaluations per second."""
    import numpy as np
    import solaraxionraytracing_amd as sa
    full, flags = make_setup("babyiaxo_xmm_ascan16")
    angles = np.linspace(0.0, 0.3, SCAN_ANGLES)
    with sa.RayTracer(full) as rt:
        rt.trace_angular_scan(angles, n, seed=2, flags=flags)   # clocks up, tables in cache (a launch of the timed size: behind a shorter
                                                                # one the first timed launch runs below the clock of the second, and the
                                                                # two-point cost model below charges the difference to t_shared)
        rt.enable_kernel_timing(True)
        t0 = time.perf_counter()
        per_angle, shared = rt.trace_angular_scan(angles, n, seed=1, flags=flags)
        wall = time.perf_counter() - t0
        ms, n_launch = rt.kernel_timing()
        # the kernel's cost model: time per ray = t_shared (sampling, bore, pipes, energy draw: once per launch) +
        # t_angle per angle; from this launch and one of 32 angles over the same range
        rt.trace_angular_scan(np.linspace(0.0, 0.3, 2 * SCAN_ANGLES), n, seed=1, flags=flags)
        ms32, n32 = rt.kernel_timing()
        pick = np.linspace(1, SCAN_ANGLES - 1, host_loop_points).astype(int)   # (not angle 0: the unrotated kernel)
        rt.set_telescope_angles(turned_y_deg=float(angles[pick[0]]))
        rt.trace_flux(50_000_000, seed=2, flags=flags)
        rt.kernel_timing()                                         # (resets: the warm-up launch is not timed)
        loop = []
        for k in pick:
            rt.set_telescope_angles(turned_y_deg=float(angles[k]))
            loop.append(rt.trace_flux(n, seed=1, flags=flags)["SUM_WEIGHTS"])
        ms_loop, n_loop = rt.kernel_timing()
        rt.set_telescope_angles(turned_y_deg=0.0)
        rt.enable_kernel_timing(False)
    assert shared["N_RAYS"] == n and n_launch == 1 and n32 == 1 and n_loop == host_loop_points
    t_angle = (ms32 - ms) * 1e9 / (SCAN_ANGLES * n)                # ps per (ray, angle)
    t_shared = ms * 1e9 / n - SCAN_ANGLES * t_angle               # ps per ray and launch
    model = sharding_model(t_shared, t_angle)
    rel = np.abs(per_angle["SUM_WEIGHTS"][pick] / np.array(loop) - 1.0).max()
    assert rel < 1e-12, rel                                        # the same rays, the same weights
    fused = n * SCAN_ANGLES / (ms / 1e3)
    host_loop = n / (ms_loop / 1e3 / n_loop)                       # (ray, angle) evaluations per second of one flux-only re-trace per angle
    avg_s = ms / 1e3 / n_launch
    summ = {"N_PASSED_TILL_WINDOW": float(per_angle["N_PASSED_TILL_WINDOW"].mean()), "N_SHELL_SELECTED": float(per_angle["N_SHELL_SELECTED"].mean())}
    return {"workload": WORKLOADS["babyiaxo_xmm_ascan16"], "ray_angle_evaluations_per_s": fused, "rays_per_s": n / (ms / 1e3),
            "ms_per_scan": ms, "launches_per_scan": n_launch, "wall_ms_per_scan": wall * 1e3,
            "host_loop_ray_angle_evaluations_per_s": host_loop, "host_loop_ms_per_angle": ms_loop / n_loop,
            "speedup_over_host_loop": fused / host_loop, "max_rel_diff_to_host_loop": float(rel),
            "passed_fraction_per_angle": [round(float(x), 5) for x in per_angle["N_PASSED"] / n],
            "ms_per_scan_32_angles": ms32, "sharding_model": model,
            "roofline": side_roofline("babyiaxo_xmm_ascan16", float(n), avg_s, n_launch, summ, float(n))}


def sharding_model(t_shared: float, t_angle: float, k: int = 50, per_launch: int = 32):
    """BASELINE configs[3] on G ranks, predicted from the fused kernel's two measured constants (ps): every launch pays t_shared per
    ray once (sampling, bore, pipes, energy draw) and t_angle per (ray, angle); `k` angles, at most `per_launch` per launch.
      by ray id   every rank turns 1/G of the rays through all k angles in L = ceil(k / per_launch) launches: (L t_shared + k t_angle) / G
      by angle    every rank turns all rays through its ceil(k / G) angles: ceil(ceil(k / G) / per_launch) t_shared + ceil(k / G) t_angle
    (ps per ray of the scan; one reduce of 8 (k + 1) slots / one all-reduce of the k-vector behind either).  By ray id is never
    slower: the difference is t_shared (launches of a rank - L / G) + t_angle (ceil(k / G) - k / G) >= 0.  A later N-GPU run can be
    held against these figures (tools/scan.py angular --fused --gpus N [--shard bins])."""
    import math
    out = {"angles": k, "angles_per_launch": per_launch, "t_shared_ps_per_ray": t_shared, "t_angle_ps_per_ray_angle": t_angle}
    launches = math.ceil(k / per_launch)
    for g in (1, 2, 4, 8):
        mine = math.ceil(k / g)
        out["by_ray_id_g%d_ps_per_ray" % g] = (launches * t_shared + k * t_angle) / g
        out["by_angle_g%d_ps_per_ray" % g] = math.ceil(mine / per_launch) * t_shared + mine * t_angle
    return out


def record_interface_block(full, n: int = 20_000_000):
    """The record-level boundary (traceAxionWrapper's own shape: 208-byte Axion records into caller memory) on the headline
    workload - PCIe-inclusive, so never `value`: every record (sart_trace_records), and only the records the reference's
    consumers keep (sart_trace_records_passed: filterIt(it.passed), raytracer.nim:2252, :2800), both into pages that exist."""
    import numpy as np
    import solaraxionraytracing_amd as sa
    from solaraxionraytracing_amd import _lib as L
    with sa.RayTracer(full) as rt:
        buf = np.zeros(n, dtype=L.AXION_DTYPE)
        rt.traceAxionWrapperPassed(200_000, seed=5)
        p = rt.trace_params(n, seed=5)
        times = {}
        for rep in range(2):                                   # the second round writes into mapped pages
            t0 = time.perf_counter()
            L.check(rt.lib.sart_trace_records(rt.handle, C.byref(p), buf.ctypes.data_as(C.c_void_p)))
            times["all"] = time.perf_counter() - t0
        n_passed = int((buf["passed"] != 0).sum(dtype=np.int64))
        check = buf[:200_000].view(np.uint8).reshape(-1, 208)[buf[:200_000]["passed"] != 0].tobytes()
        for rep in range(2):
            t0 = time.perf_counter()
            got, cnt = rt.traceAxionWrapperPassed(n, seed=5, out=buf)
            times["passed"] = time.perf_counter() - t0
        assert cnt["n_passed"] == n_passed == len(got) and got[:len(check) // 208].tobytes() == check
    return {"workload": "record interface, BabyIAXO / XMM: 208-byte Axion records into caller memory (PCIe-inclusive; mapped pages)",
            "rays": n, "passed_fraction": n_passed / n,
            "every_record": {"rays_per_s": n / times["all"], "gb_per_s": n * 208 / times["all"] / 1e9},
            "passed_records_only": {"rays_per_s": n / times["passed"], "gb_per_s": n_passed * 208 / times["passed"] / 1e9,
                                    "byte_identical_to_filtered_buffer": True}}


def fixed64_block(full, f64_rate: float, n: int = 1_000_000_000, launches: int = 10):
    """The deterministic accumulation mode (SART_ACCUM_FIXED64: integer atomics, int64 reduce) on the headline workload:
    its rate beside the f64 rate of the line, the distance between the two images, and a bitwise check (the same rays as
    one launch and as three launches into another replica layout)."""
    import numpy as np
    import solaraxionraytracing_amd as sa
    with sa.RayTracer(full) as rt:
        rt.set_accumulation_mode("fixed64")
        for k in range(3):
            rt.trace_histogram(n, seed=2, ray_id_offset=k * n)
        rt.enable_kernel_timing(True)
        for k in range(launches):
            img, s = rt.trace_histogram(n, seed=1, ray_id_offset=k * n, accumulate=(k > 0))
        ms, n_launch = rt.kernel_timing()
        rt.enable_kernel_timing(False)
        m = 100_000_000
        img_a, s_a = rt.trace_histogram(m, seed=3)
        q = rt.fixed_quanta()
        rt.set_accumulation_mode("f64")
        img_f, s_f = rt.trace_histogram(m, seed=3)
    os.environ["SART_IMAGE_REPLICAS"] = "32"
    try:
        with sa.RayTracer(full) as rt:
            rt.set_accumulation_mode("fixed64")
            for k, (lo, hi) in enumerate(((0, 1), (1, 33_333_334), (33_333_334, m))):
                img_b, s_b = rt.trace_histogram(hi - lo, seed=3, ray_id_offset=lo, accumulate=(k > 0))
    finally:
        del os.environ["SART_IMAGE_REPLICAS"]
    assert s["N_RAYS"] == float(n) * launches and abs(img.sum() - s["SUM_WEIGHTS"]) <= 1e-12 * s["SUM_WEIGHTS"]
    rate = n * n_launch / (ms / 1e3)
    return {"mode": "SART_ACCUM_FIXED64", "rays_per_s": rate, "ms_per_launch": ms / n_launch, "rays_per_launch": n,
            "vs_f64_rate": rate / f64_rate, "weight_quantum": q["weight"],
            "bitwise_equal_across_launch_split_and_replicas": bool(np.array_equal(img_a.view(np.uint64), img_b.view(np.uint64)) and
                                                                   s_a["SUM_WEIGHTS"] == s_b["SUM_WEIGHTS"]),
            "max_abs_diff_to_f64_image_over_peak": float(np.abs(img_a - img_f).max() / img_f.max()),
            "rel_diff_sum_weights_to_f64": float(abs(s_a["SUM_WEIGHTS"] / s_f["SUM_WEIGHTS"] - 1.0))}


def emission_table_rate(reps: int = 5):
    """The second kernel of the library: the solar emission table of readOpacityFile.nim's cell loop,
    1968 radii x 1500 energies, all eight terms.  Unit = one (radius, energy) cell; roofline from its own PMC profile
    (profiles/pmc_current.json["emission_table"]: f64 flop per cell) x cells / live HIP-event kernel time."""
    import numpy as np
    import solaraxionraytracing_amd.emission as em
    from solaraxionraytracing_amd import tables
    zones = em.solar_zones()
    _, energies = tables.solar_grid()
    cells = len(zones) * energies.size
    em.emission_table(zones, energies)
    ms = []
    for _ in range(reps):
        table = em.emission_table(zones, energies)
        ms.append(em.last_kernel_ms())
    assert np.all(np.isfinite(table)) and table.max() > 0
    kernel_s = float(np.median(ms)) * 1e-3
    blk = {"bound": "f64-valu-issue", "achieved": None, "peak": F64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": None, "traffic": None,
           "kernel": "emission_table_kernel", "avg_kernel_ms": kernel_s * 1e3, "cells_per_launch": cells}
    pmc, why_not, lib_id = pmc_for_this_build("emission_table")
    blk["build_id"] = lib_id
    if pmc is None:
        blk["note"] = why_not
    else:
        tf = pmc["f64_flop_per_ray"] * cells / kernel_s / 1e12          # "ray" = unit of the profiled launch = one cell
        blk.update({"achieved": tf, "frac": tf / F64_VALU_PEAK_TFLOPS, "f64_flop_per_cell": pmc["f64_flop_per_ray"],
                    "valu_issue_utilisation": pmc.get("valu_issue_utilisation"), "pmc_source": pmc.get("source")})
        assert blk["frac"] <= 1.0 and (blk["valu_issue_utilisation"] is None or blk["valu_issue_utilisation"] <= 1.0), blk
    return {"workload": "solar emission table, AGSS09 model, 1968 radii x 1500 energies, eight terms (readOpacityFile.nim:745-860)",
            "cells_per_s": cells / kernel_s, "ms_per_launch": kernel_s * 1e3, "roofline": blk}


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
    # the same curve through the fused scan kernel (every angle on the ray ids [0, rays_per_angle)), against the oracle on those ids
    with sa.RayTracer(full) as rt:
        _, _, fused_rel = sa.performAngularScan(rt, 0, 0, 1, rays_per_angle, flags=flags, angles=angles, fused=True)
    cpu_f = np.empty(points)
    for i, a in enumerate(angles):
        s = full.setup.copy()
        s.telescope_turned_y_deg = float(a)
        cpu_f[i] = o.trace_histogram(rays_per_angle, flags=flags, setup=s, n_threads=available_cpus())[1]["SUM_WEIGHTS"]
    cpu_f_rel = cpu_f / cpu_f.max()
    return {"value": float(np.sqrt(np.mean((gpu_rel - cpu_rel) ** 2))), "points": points, "rays_per_angle": rays_per_angle,
            "gpu_relative_flux": [round(float(x), 6) for x in gpu_rel], "cpu_relative_flux": [round(float(x), 6) for x in cpu_rel],
            "fused_scan_value": float(np.sqrt(np.mean((fused_rel - cpu_f_rel) ** 2))),
            "fused_scan_gpu_relative_flux": [round(float(x), 6) for x in fused_rel],
            "fused_scan_cpu_relative_flux": [round(float(x), 6) for x in cpu_f_rel]}


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
    bounded sample of the same workload.  Timed with the -O3 -march=native build (oracle/libsart_oracle_native.so, SURVEY
    8(d)); the parity tests use the portable -O2 build."""
    from oracle.oracle import Oracle
    cores = available_cpus()
    try:
        o = Oracle(full, variant="native")
        build = "gcc -O3 -march=native -fopenmp -ffp-contract=off"
    except Exception:
        o = Oracle(full)
        build = "gcc -O2 -fopenmp -ffp-contract=off"
    o.trace_histogram(200_000, seed=seed, n_threads=cores)   # warm up threads / page in tables
    t0 = time.perf_counter()
    _, summ, used = o.trace_histogram(n_sample, seed=seed, n_threads=cores)
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    n1 = max(200_000, n_sample // 50)
    o.trace_histogram(n1, seed=seed, n_threads=1)
    dt1 = time.perf_counter() - t0
    return {"value": n_sample / dt, "unit": "rays/s", "cores": used, "kind": "port", "single_thread_rays_per_s": n1 / dt1,
            "sample": "%d rays of the same workload, C restatement of traceAxion (oracle/sart_oracle.c, %s), "
                      "%.1f s; single thread: %.3g rays/s" % (n_sample, build, dt, n1 / dt1)}


if __name__ == "__main__":
    main()
