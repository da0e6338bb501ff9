#!/usr/bin/env python3
"""Summarises the per-dispatch PMC CSVs of tools/pmc_profile.sh for one kernel: mean counter value per launch."""
import csv, glob, os, sys, json
d = sys.argv[1]
kernel = sys.argv[2] if len(sys.argv) > 2 else "trace_histogram"
res = {}
for f in sorted(glob.glob(os.path.join(d, "pass*", "**", "*counter_collection.csv"), recursive=True)):
    acc = {}
    for r in csv.DictReader(open(f)):
        if kernel not in r["Kernel_Name"]:
            continue
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in acc.items():
        res[k] = sum(v) / len(v)
for k in sorted(res):
    print("%-28s %.6g" % (k, res[k]))
if "SQ_WAVE_CYCLES" in res:
    w = res
    g = lambda k: w.get(k, float("nan"))
    print("--- derived (per launch)")
    print("VALU insts / wave            %.1f" % (g("SQ_INSTS_VALU") / g("SQ_WAVES")))
    print("VALU lane utilisation        %.3f" % (g("SQ_THREAD_CYCLES_VALU") / (64.0 * g("SQ_ACTIVE_INST_VALU"))))
    print("wave-cycles/wave (x4 = clk)  %.0f" % (g("SQ_WAVE_CYCLES") / g("SQ_WAVES")))
    print("WAIT_ANY / WAVE_CYCLES       %.3f" % (g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")))
    print("WAIT_INST_ANY / WAVE_CYCLES  %.3f" % (g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES")))
    print("ACTIVE_INST_VALU / WAVE_CYC  %.3f" % (g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES")))
json.dump(res, open(os.path.join(d, "summary.json"), "w"), indent=1)
