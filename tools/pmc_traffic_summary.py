#!/usr/bin/env python3
"""Turns the counters of tools/pmc_traffic.sh (gpurun_out/pmct_<tag>) into profiles/pmc_traffic.json, the per-ray fabric
traffic bench.py scales into `roofline.traffic`.  Usage: python tools/pmc_traffic_summary.py gpurun_out/pmct_<tag> [rays]"""
import csv, glob, json, os, sys
d = sys.argv[1]
rays = float(sys.argv[2]) if len(sys.argv) > 2 else 1e8
res = {}
for f in sorted(glob.glob(os.path.join(d, "pass*", "**", "*counter_collection.csv"), recursive=True)):
    acc = {}
    for r in csv.DictReader(open(f)):
        if "trace_histogram" in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in acc.items():
        res[k] = sum(v) / len(v)
rd_by_size = 32 * res["TCC_EA0_RDREQ_32B_sum"] + 64 * res["TCC_EA0_RDREQ_64B_sum"] + 128 * res["TCC_EA0_RDREQ_128B_sum"]
rd_fetch = 2.0 * res["FETCH_SIZE"] * 1024.0          # gfx950: FETCH_SIZE (KiB) under-reports by 2x (microarch guide)
wr = res["WRITE_SIZE"] * 1024.0
out = {"workload": "babyiaxo_xmm", "bytes_per_ray": (rd_by_size + wr) / rays, "read_bytes_per_ray": rd_by_size / rays,
       "read_bytes_per_ray_from_2x_fetch_size": rd_fetch / rays, "write_bytes_per_ray": wr / rays,
       "rays_per_launch_measured": rays, "counters": res,
       "note": "rocprofv3 --pmc in separate passes (tools/pmc_traffic.sh) on 1e8-ray launches, scaled linearly to this launch size. "
               "Reads: TCC_EA0_RDREQ by request size (all 128-B lines here), which equals 2 x FETCH_SIZE x 1024 - the gfx950 FETCH_SIZE "
               "x2 correction of the microarch guide; writes: WRITE_SIZE x 1024 (plain 64-B writes + f64 atomics tallied at 32 B). "
               "Fabric-side counters: Infinity-Cache hits are included, so this is L2<->fabric traffic (the 44 MB of tables stay resident "
               "in the 256 MB Infinity Cache), an upper bound on HBM traffic."}
json.dump(out, open(os.path.join(d, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k: out[k] for k in ("bytes_per_ray", "read_bytes_per_ray", "read_bytes_per_ray_from_2x_fetch_size", "write_bytes_per_ray")}))
