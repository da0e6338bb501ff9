#!/usr/bin/env python3
"""Summarises the per-dispatch PMC CSVs of tools/pmc_profile.sh for the hot-path kernel: mean counter value per launch
plus the derived per-ray figures `bench.py` turns into its `roofline` block.

  python tools/pmc_summary.py gpurun_out/pmc_<tag> --workload babyiaxo_xmm --rays 1e9 [--publish r02_v16]

--publish NAME copies the summary to profiles/NAME_<workload>_pmc_summary.json and points profiles/pmc_current.json's entry
for <workload> at it (bench.py reads pmc_current.json; it cannot collect PMC counters from inside its own process).

Derived figures and their formulas (all per launch of `rays` rays, counters are means over the profiled launches):
  valu_insts_per_64_rays  = SQ_INSTS_VALU * 64 / rays
  f64_flop_per_ray        = (2 FMA_F64 + MUL_F64 + ADD_F64 + TRANS_F64) * 64 lanes / rays      (wave instructions x 64 lanes:
                            what the vector unit ISSUES; lanes switched off by EXEC are included)
  f64_flop_per_ray_active_lanes = f64_flop_per_ray x valu_lane_utilisation   (SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU):
                            the share of issued lanes that are enabled, taken over ALL vector instructions - the f64 ones
                            are assumed to have the average)
  valu_issue_utilisation  = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES * waves_per_simd, waves_per_simd = SQ_WAVES / (4 n_cu)
                            (both counters are quad-cycles summed over waves; every wave lives for the whole launch);
                            cross-check / short-lived waves: 4 SQ_ACTIVE_INST_VALU / (4 n_cu x GRBM_GUI_ACTIVE / 8)
  fabric_read_bytes       = 32 RDREQ_32B + 64 RDREQ_64B + 128 RDREQ_128B   (TCC_EA0_RDREQ by size; cross-check: 2 x FETCH_SIZE
                            x 1024, the gfx950 FETCH_SIZE x2 correction of MI355X_MICROARCH.md)
  fabric_write_bytes      = WRITE_SIZE x 1024
  fabric_bytes_per_ray    = (read + write) / rays        (L2 <-> fabric; Infinity-Cache hits included: upper bound on HBM)
"""
import argparse
import csv
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def collect(d, kernel):
    res = {}
    for f in sorted(glob.glob(os.path.join(d, "pass*", "**", "*counter_collection.csv"), recursive=True)):
        acc = {}
        for r in csv.DictReader(open(f)):
            if kernel not in r["Kernel_Name"]:
                continue
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        # Dispatches of the same kernel that are not the profiled launches (the 2e5-ray pilot launch that places the LDS image
        # tile, warm-up launches of another size) are recognised by their SQ_WAVES / much smaller counts and dropped: keep the
        # dispatches whose value is at least 0.8 of the largest one.
        for k, v in acc.items():
            top = max(v)
            keep = [x for x in v if x >= 0.8 * top] if top > 0 else v
            res[k] = sum(keep) / len(keep)
    return res


def derive(w, rays, n_cu):
    g = lambda k: w.get(k, float("nan"))
    out = {}
    if "SQ_WAVE_CYCLES" in w:
        wps = g("SQ_WAVES") / (4.0 * n_cu)
        out["waves_per_simd"] = wps
        out["valu_insts_per_64_rays"] = g("SQ_INSTS_VALU") * 64.0 / rays
        out["valu_issue_utilisation"] = g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES") * wps
        out["valu_lane_utilisation"] = g("SQ_THREAD_CYCLES_VALU") / (64.0 * g("SQ_ACTIVE_INST_VALU"))
        out["wait_any_fraction"] = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")
        out["wait_inst_any_fraction"] = g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES")
        out["cycles_per_valu_inst"] = 4.0 * g("SQ_ACTIVE_INST_VALU") / g("SQ_INSTS_VALU")
    if "SQ_INSTS_VALU_FMA_F64" in w:
        f64 = g("SQ_INSTS_VALU_FMA_F64") + g("SQ_INSTS_VALU_MUL_F64") + g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_TRANS_F64")
        out["f64_flop_per_ray"] = (2.0 * g("SQ_INSTS_VALU_FMA_F64") + g("SQ_INSTS_VALU_MUL_F64") + g("SQ_INSTS_VALU_ADD_F64") +
                                   g("SQ_INSTS_VALU_TRANS_F64")) * 64.0 / rays
        if "valu_lane_utilisation" in out:
            out["f64_flop_per_ray_active_lanes"] = out["f64_flop_per_ray"] * out["valu_lane_utilisation"]
        if "SQ_INSTS_VALU" in w:
            out["f64_share_of_valu_insts"] = f64 / g("SQ_INSTS_VALU")
            out["int_share_of_valu_insts"] = (g("SQ_INSTS_VALU_INT32") + g("SQ_INSTS_VALU_INT64")) / g("SQ_INSTS_VALU")
    if "SQ_LDS_BANK_CONFLICT" in w:
        out["lds_conflict_cycles_per_active_lds_cycle"] = g("SQ_LDS_BANK_CONFLICT") / max(1.0, g("SQ_ACTIVE_INST_LDS"))
    if "TCC_EA0_RDREQ_128B_sum" in w:
        rd = 32 * g("TCC_EA0_RDREQ_32B_sum") + 64 * g("TCC_EA0_RDREQ_64B_sum") + 128 * g("TCC_EA0_RDREQ_128B_sum")
        out["fabric_read_bytes_per_ray"] = rd / rays
        if "FETCH_SIZE" in w:
            out["fabric_read_bytes_per_ray_from_2x_fetch_size"] = 2.0 * g("FETCH_SIZE") * 1024.0 / rays
        if "WRITE_SIZE" in w:
            out["fabric_write_bytes_per_ray"] = g("WRITE_SIZE") * 1024.0 / rays
            out["fabric_bytes_per_ray"] = out["fabric_read_bytes_per_ray"] + out["fabric_write_bytes_per_ray"]
    if "GRBM_GUI_ACTIVE" in w and "SQ_ACTIVE_INST_VALU" in w:
        # same quantity from the clock counter: VALU-active cycles summed over waves / (SIMDs x kernel cycles), kernel cycles =
        # GRBM_GUI_ACTIVE / 8 (rocprofv3 reports the sum over the 8 XCDs).  Valid for kernels whose waves do not live for the
        # whole launch too (the emission kernel), where the per-wave form above over-counts.
        out["valu_issue_utilisation_from_clock"] = 4.0 * g("SQ_ACTIVE_INST_VALU") / (4.0 * n_cu * g("GRBM_GUI_ACTIVE") / 8.0)
        if out.get("waves_per_simd", 0.0) > 8.0:
            out["valu_issue_utilisation"] = out["valu_issue_utilisation_from_clock"]
    if "TCC_HIT_sum" in w:
        out["l2_hit_rate"] = g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))
    return out


def library_build_id(profile_dir):
    """The build the counters were collected on: tools/pmc_profile.sh writes sart_build_id() of the library it profiled to
    <dir>/build_id.txt on the GPU box; without that file, the libsart.so of this tree (loading it needs no GPU)."""
    try:
        return open(os.path.join(profile_dir, "build_id.txt")).read().strip()
    except OSError:
        pass
    import sys
    sys.path.insert(0, ROOT)
    from solaraxionraytracing_amd import _lib
    return _lib.build_id()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--kernel", default="trace_histogram")
    ap.add_argument("--workload", default="babyiaxo_xmm")
    ap.add_argument("--rays", type=float, default=1e9, help="rays per profiled launch (bench.py --rays-per-step)")
    ap.add_argument("--n-cu", type=int, default=256)
    ap.add_argument("--publish", default=None)
    args = ap.parse_args()
    res = collect(args.dir, args.kernel)
    der = derive(res, args.rays, args.n_cu)
    for k in sorted(res):
        print("%-32s %.6g" % (k, res[k]))
    print("--- derived")
    for k, v in der.items():
        print("%-48s %.6g" % (k, v))
    build = library_build_id(args.dir)
    out = {"workload": args.workload, "kernel": args.kernel, "rays_per_launch": args.rays, "n_cu": args.n_cu, "build_id": build,
           "derived": der, "counters": res,
           "how": "rocprofv3 --pmc in separate passes (tools/pmc_profile.sh), mean over the profiled launches; formulas in "
                  "tools/pmc_summary.py"}
    json.dump(out, open(os.path.join(args.dir, "summary.json"), "w"), indent=1)
    if args.publish:
        name = "%s_%s_pmc_summary.json" % (args.publish, args.workload)
        json.dump(out, open(os.path.join(ROOT, "profiles", name), "w"), indent=1)
        cur_path = os.path.join(ROOT, "profiles", "pmc_current.json")
        try:
            cur = json.load(open(cur_path))
        except Exception:
            cur = {}
        cur[args.workload] = {"source": "profiles/" + name, "rays_per_launch": args.rays, "build_id": build, **der}
        json.dump(cur, open(cur_path, "w"), indent=1)
        print("published profiles/%s" % name)


if __name__ == "__main__":
    main()
