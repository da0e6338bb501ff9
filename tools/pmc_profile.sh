#!/bin/bash
# Collects rocprofv3 PMC counters for the hot-path kernel in separate passes (never combined with
# tracing, see the gpurun rules).  Usage on the GPU box:  bash tools/pmc_profile.sh <tag> [bench args...]
# Output: gpurun_out/pmc_<tag>/pass*/...counter_collection.csv ; summarise with tools/pmc_summary.py.
# Passes 1-3: SQ (instruction mix, issue, waits, LDS); 4-9: TCC (fabric traffic by request size, FETCH_SIZE /
# WRITE_SIZE, L2 hit rate); 10: GRBM (clock).
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
# the build the counters belong to (tools/pmc_summary.py stores it; bench.py compares it with the library it times)
python3 -c "import sys; sys.path.insert(0, '$ROOT'); from solaraxionraytracing_amd import _lib; print(_lib.build_id())" > $OUT/build_id.txt
cd /tmp && export TMPDIR=/tmp
PASSES=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"
 "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_SALU SQ_INSTS_LDS"
 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_CVT"
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
 "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_ATOMIC_sum TCC_EA0_RDREQ_DRAM_sum"
 "FETCH_SIZE"
 "WRITE_SIZE TCC_EA0_WRREQ_DRAM_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum"
 "GRBM_GUI_ACTIVE GRBM_COUNT"
)
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  if [ -n "$PMC_PASSES" ] && ! echo " $PMC_PASSES " | grep -q " $i "; then continue; fi   # PMC_PASSES="1 2": only those passes
  if [ -n "$PMC_PROGRAM" ]; then   # another program of this repository under the same passes (e.g. tools/emission_bench.py --no-cpu)
    rocprofv3 --pmc $P --output-format csv -d $OUT/pass$i -- python3 $ROOT/$PMC_PROGRAM "$@" > $OUT/pass$i.log 2>&1 || { echo "pass $i FAILED ($P)" >&2; tail -5 $OUT/pass$i.log >&2; exit 1; }
  else
    rocprofv3 --pmc $P --output-format csv -d $OUT/pass$i -- python3 $ROOT/bench.py --profile-run --no-proof --steps 2 --warmup 1 "$@" > $OUT/pass$i.log 2>&1 || { echo "pass $i FAILED ($P)" >&2; tail -5 $OUT/pass$i.log >&2; exit 1; }
  fi
  echo "pass $i done: $P"
done
