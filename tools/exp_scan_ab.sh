#!/bin/bash
# Same-box A/B of library builds on the two fused-scan workloads (bench.py --profile-run): bash tools/exp_scan_ab.sh libA.so libB.so ...
for R in 1 2; do
for L in "$@"; do
  if [ "$L" = default ]; then unset SART_LIBSART; else export SART_LIBSART=$PWD/$L; fi
  python bench.py --workload babyiaxo_xmm_gas_scan32 --profile-run --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', 'scan32 ms/step', round(d['ms_per_step'],3), 'flux at resonance', repr(d['results']['flux']))"
  python bench.py --workload babyiaxo_xmm_gas_scan32 --profile-run --steps 6 --warmup 2 --accumulation fixed64 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', 'scan32 fixed64 ms/step', round(d['ms_per_step'],3), 'flux at resonance', repr(d['results']['flux']))"
done
done
