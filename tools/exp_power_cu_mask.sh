#!/bin/bash
# Socket power, shader clock and rate of the ray kernel on all 256 CUs and on a part of them (HSA_CU_MASK, per process; no
# machine setting is touched): is the full-chip rate held by the power / current management?  Usage on the GPU box:
#   bash tools/exp_power_cu_mask.sh > gpurun_out/power_cu_mask.txt
# The persistent kernel starts one workgroup per CU it is told about (256): under a mask of n CUs they run in 256 / n rounds,
# so masks that divide 256 keep the work per CU equal.
export POWER_SECONDS=${POWER_SECONDS:-12} POWER_DELAY=${POWER_DELAY:-7} POWER_SAMPLES=${POWER_SAMPLES:-4}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
echo "columns: junction temperature (C), shader clock, socket power (W); then the rate of tools/power_run.py"
for W in babyiaxo_xmm cast_llnl_gold; do
  echo "== $W, all CUs";                   bash $ROOT/tools/power_sample.sh --workload $W
  echo "== $W, HSA_CU_MASK=0:0-127 (128 CUs)"; HSA_CU_MASK=0:0-127 bash $ROOT/tools/power_sample.sh --workload $W
  echo "== $W, HSA_CU_MASK=0:0-63 (64 CUs)";   HSA_CU_MASK=0:0-63 bash $ROOT/tools/power_sample.sh --workload $W
done
