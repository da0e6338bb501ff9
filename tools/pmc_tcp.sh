#!/bin/bash
# Vector-memory path counters of the ray kernel (TA / TCP / UTCL1 / TCP->TCC latency), separate rocprofv3 --pmc passes.
# Usage on the GPU box:  bash tools/pmc_tcp.sh <tag> [bench args...]   -> gpurun_out/pmc_tcp_<tag>/
# The four TA counters of round 2's pass 4 do not fit one pass (rocprofv3 aborted with signal 6, "Request exceeds the
# capabilities of the hardware", gpurun_out/pmc_tcp_v23/pass4.log): they are two passes now, and a pass that fails ends
# the script with its exit code instead of being skipped.  Round 3: the TD pass (three TD counters + one TCP counter) aborts the
# same way and is split as well.
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_tcp_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PASSES=(
 "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum"
 "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum"
 "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum"
 "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum"
 "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum"
 "TD_TD_BUSY_sum TD_TC_STALL_sum"
 "TD_LOAD_WAVEFRONT_sum TCP_TD_TCP_STALL_CYCLES_sum"
)
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  if [ -n "$PMC_PASSES" ] && ! echo " $PMC_PASSES " | grep -q " $i "; then continue; fi
  if ! timeout -k 5 150 rocprofv3 --pmc $P --output-format csv -d $OUT/pass$i -- python3 $ROOT/bench.py --profile-run --no-proof --steps 2 --warmup 1 "$@" > $OUT/pass$i.log 2>&1; then
    echo "pass $i FAILED ($P): tail of $OUT/pass$i.log" >&2
    tail -5 $OUT/pass$i.log >&2
    exit 1
  fi
  echo "pass $i done: $P"
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(list)
for f in glob.glob("$OUT/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "trace_histogram" in r["Kernel_Name"]:
            tot[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(tot):
    v = tot[k]
    print("%-48s %.6g (mean of %d dispatches)" % (k, sum(v) / len(v), len(v)))
PY
