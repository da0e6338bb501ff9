#!/bin/bash
# Fabric bytes per ray of CAST / LLNL with one gather table at a time folded onto a few rows (experiment build, wrong results by
# design): what each table contributes to the L2-miss traffic.  Run on the GPU box; summaries: tools/pmc_summary.py.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export SART_LIBSART=$ROOT/tools/microbench/libsart_dbg.so
for F in 00000000 04000000 02000000 01000000 07000000; do
  SART_DEBUG_FLAGS=$F PMC_PASSES="4 7" bash $ROOT/tools/pmc_profile.sh r03_bytes_$F --workload cast_llnl_gold --rays-per-step 1e8 > /dev/null
  python3 $ROOT/tools/pmc_summary.py $ROOT/gpurun_out/pmc_r03_bytes_$F --workload cast_llnl_gold --rays 1e8 | grep -E "fabric_(read_)?bytes_per_ray " | sed "s/^/flags $F  /"
done
