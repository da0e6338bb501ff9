for L in tools/microbench/libsart_F.so default tools/microbench/libsart_K.so; do
  if [ "$L" = default ]; then unset SART_LIBSART; else export SART_LIBSART=$PWD/$L; fi
  for R in 1 2; do
  python bench.py --workload babyiaxo_xmm_gas_scan32 --profile-run --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', 'scan32 ms/step', round(d['ms_per_step'],3))"
  python bench.py --workload babyiaxo_xmm_ascan16 --profile-run --steps 10 --warmup 2 --rays-per-step 2e8 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', 'ascan16 ms/step', round(d['ms_per_step'],3))"
  done
done
