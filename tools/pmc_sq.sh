#!/bin/bash
# Sequencer-side counters of the ray kernel (instruction fetch, scalar unit, in-flight levels = latencies, FIFO stalls),
# separate rocprofv3 --pmc passes.  Usage on the GPU box:  bash tools/pmc_sq.sh <tag> [bench args...] -> gpurun_out/pmc_sq_<tag>/
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PASSES=(
 "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_WAVE_CYCLES"
 "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"
 "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_STALL SQC_ICACHE_BUSY_CYCLES"
 "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_IDX_ACTIVE SQ_CYCLES"
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
    print("%-36s %.6g (mean of %d dispatches)" % (k, sum(v) / len(v), len(v)))
PY
