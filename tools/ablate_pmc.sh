#!/bin/bash
# Stage ablation of the ray kernel: VALU instruction counts and time for A0 only / A0+A1 / full (builds under variants/).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in a0 a0a1 full; do
  export SART_LIBSART=$ROOT/solaraxionraytracing_amd/variants/libsart_$v.so
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 --output-format csv -d $ROOT/gpurun_out/ablate_$v/pass1 -- python3 $ROOT/bench.py --profile-run --steps 2 --warmup 1 --rays-per-step 1e8 > $ROOT/gpurun_out/ablate_$v.log 2>&1
  python3 $ROOT/tools/pmc_summary.py $ROOT/gpurun_out/ablate_$v | grep -E "SQ_INSTS|VALU insts|ACTIVE_INST_VALU /"
  python3 $ROOT/bench.py --profile-run --steps 5 --warmup 2 --rays-per-step 1e8 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', 'ms/1e8', d['roofline']['avg_kernel_ms'])"
done
