#!/bin/bash
# Fabric (L2 -> Infinity Cache / HBM) request counters of the hot-path kernel by request size, to turn FETCH_SIZE /
# WRITE_SIZE into bytes for THIS access pattern (the microarch guide's x2 FETCH_SIZE correction is calibrated for wide
# coalesced streams only).  Usage on the GPU box: bash tools/pmc_traffic.sh <tag> [bench args]
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmct_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PASSES=(
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
 "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_ATOMIC_sum TCC_EA0_RDREQ_DRAM_sum"
 "FETCH_SIZE"
 "WRITE_SIZE TCC_EA0_WRREQ_DRAM_sum"
)
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $P --output-format csv -d $OUT/pass$i -- python3 $ROOT/bench.py --profile-run --steps 2 --warmup 1 --rays-per-step 1e8 "$@" > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
  echo "pass $i done: $P"
done
