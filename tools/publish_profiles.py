#!/usr/bin/env python3
"""Copies what one measurement round (tools/gpu_round.sh <tag> ...) left under gpurun_out/ into profiles/ under the tag's name and
re-publishes profiles/pmc_current.json:  python tools/publish_profiles.py r05_v55

  gpurun_out/pmc_<tag>_<workload>/       -> profiles/<tag>_<workload>_pmc_summary.json (tools/pmc_summary.py --publish)
  gpurun_out/bench_<tag>.json            -> profiles/<tag>_bench.json
  gpurun_out/prof_<tag>/**kernel_stats   -> profiles/<tag>_kernel_stats.csv + <tag>_kernel_trace_summary.json (per-dispatch durations)
  gpurun_out/prof_<tag>_scan/**          -> profiles/<tag>_scan32_kernel_stats.csv
  gpurun_out/<tag>_*.{json,md,txt}       -> profiles/ (full-size comparisons, throughput table, sustained runs, sequencer counters)"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
PROF = os.path.join(ROOT, "profiles")
WORKLOADS = {   # workload -> (rays per profiled launch, kernel-name substring)
    "babyiaxo_xmm": (1e9, "trace_histogram"), "cast_llnl_gold": (1e8, "trace_histogram"), "cast_llnl": (1e8, "trace_histogram"), "babyiaxo_xmm_gas": (1e8, "trace_histogram"),
    "babyiaxo_xmm_rot": (1e8, "trace_histogram"), "babyiaxo_xmm_gas_scan32": (1e9, "trace_histogram"),
    "babyiaxo_xmm_ascan16": (2e8, "trace_angular_scan"), "emission_table": (2952000.0, "emission_table_kernel"),
}


def main():
    tag = sys.argv[1]
    for w, (rays, kernel) in WORKLOADS.items():
        d = os.path.join(OUT, "pmc_%s_%s" % (tag, w))
        if os.path.isdir(d):
            subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), d, "--workload", w, "--rays", repr(rays), "--kernel", kernel,
                            "--publish", tag], check=True, capture_output=True)
            print("published", w)
    b = os.path.join(OUT, "bench_%s.json" % tag)
    if os.path.exists(b):
        shutil.copy(b, os.path.join(PROF, "%s_bench.json" % tag))
    for sub, name in (("prof_%s" % tag, "%s_kernel_stats.csv" % tag), ("prof_%s_scan" % tag, "%s_scan32_kernel_stats.csv" % tag)):
        stats = glob.glob(os.path.join(OUT, sub, "**", "*kernel_stats.csv"), recursive=True)
        if stats:
            shutil.copy(stats[0], os.path.join(PROF, name))
    trace = glob.glob(os.path.join(OUT, "prof_%s" % tag, "**", "*kernel_trace.csv"), recursive=True)
    if trace:
        rows = [r for r in csv.DictReader(open(trace[0])) if "trace_histogram_kernel" in r["Kernel_Name"]]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
        big = [x for x in dur if x > 1.0]
        bench = json.load(open(b)) if os.path.exists(b) else {}
        out = {"command": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --profile-run --no-proof --steps 5 --warmup 2",
               "build_id": bench.get("roofline", {}).get("build_id"), "kernel": rows[0]["Kernel_Name"][:100] if rows else None,
               "dispatch_durations_ms": [round(x, 4) for x in dur], "mean_ms_of_1e9_ray_dispatches": sum(big) / max(1, len(big)),
               "mean_ms_of_last_five": sum(big[-5:]) / max(1, len(big[-5:])),
               "lds_bytes": int(rows[0].get("LDS_Block_Size", 0) or 0) if rows else None,
               "scratch_bytes": int(rows[0].get("Scratch_Size", 0) or 0) if rows else None,
               "note": "the first dispatch is the 2e5-ray pilot launch that places the LDS image tile; the --stats average over all of them is "
                       "therefore lower than the per-step time.  Bench line of the same build: profiles/%s_bench.json (%.2f ms per step)"
                       % (tag, bench.get("ms_per_step", float("nan")))}
        json.dump(out, open(os.path.join(PROF, "%s_kernel_trace_summary.json" % tag), "w"), indent=1)
    for f in glob.glob(os.path.join(OUT, "%s_*" % tag)):
        if f.endswith((".json", ".md", ".txt")) and os.path.getsize(f) > 0:
            shutil.copy(f, os.path.join(PROF, os.path.basename(f)))
    print(sorted(os.path.basename(f) for f in glob.glob(os.path.join(PROF, "%s_*" % tag))))


if __name__ == "__main__":
    main()
