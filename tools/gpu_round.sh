#!/bin/bash
# One measurement round on the GPU box:  bash tools/gpu_round.sh <tag> [tests] [bench] [stats] [pmc] [pmc_side] [pmc_scan] [pmc_ascan] [pmc_emission] [sq] [fullsize] [sustained]
#   tests     pytest -m gpu                                  -> gpurun_out/pytest_gpu_<tag>.log
#   bench     python bench.py                                -> gpurun_out/bench_<tag>.json
#   stats     rocprofv3 --kernel-trace --stats on bench.py   -> gpurun_out/prof_<tag>/
#   pmc       PMC passes, headline workload (1e9-ray launches) -> gpurun_out/pmc_<tag>_babyiaxo_xmm/
#   pmc_side  PMC passes, CAST / gas / rotated (1e8-ray launches)
# Steps are joined with && semantics (set -e): nothing runs after a failed GPU step.
set -e -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
for STEP in "$@"; do
  case $STEP in
    tests)
      (cd $ROOT && timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_$TAG.log 2>&1) || { tail -30 $ROOT/gpurun_out/pytest_gpu_$TAG.log; exit 1; }
      tail -3 $ROOT/gpurun_out/pytest_gpu_$TAG.log ;;
    bench)
      (cd $ROOT && timeout -k 10 600 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err) || { tail -20 $ROOT/gpurun_out/bench_$TAG.err; exit 1; }
      cat $ROOT/gpurun_out/bench_$TAG.json | python3 -c "import json,sys; d=json.load(sys.stdin); print(d['value'], d['ms_per_step'], json.dumps(d['roofline'])[:600]); [print(w['workload'][:40], w.get('rays_per_s', w.get('cells_per_s'))) for w in d.get('other_workloads',[])]; print(d.get('cpu_baseline'))" ;;
    stats)
      (cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_$TAG -- python3 $ROOT/bench.py --profile-run --no-proof --steps 5 --warmup 2 > $ROOT/gpurun_out/prof_$TAG.log 2>&1) || { tail -20 $ROOT/gpurun_out/prof_$TAG.log; exit 1; }
      find $ROOT/gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1 | xargs head -5 ;;
    pmc)
      (cd $ROOT && timeout -k 10 900 bash tools/pmc_profile.sh ${TAG}_babyiaxo_xmm --workload babyiaxo_xmm) ;;
    pmc_side)
      for W in cast_llnl_gold cast_llnl babyiaxo_xmm_gas babyiaxo_xmm_rot; do
        (cd $ROOT && timeout -k 10 900 bash tools/pmc_profile.sh ${TAG}_$W --workload $W --rays-per-step 1e8)
      done ;;
    pmc_emission)
      (cd $ROOT && PMC_PROGRAM=tools/emission_bench.py PMC_PASSES="1 2 3 9" timeout -k 10 600 bash tools/pmc_profile.sh ${TAG}_emission_table --no-cpu) ;;
    pmc_scan)   # the fused 32-mass scan workload (1e9-ray launches)
      (cd $ROOT && timeout -k 10 900 bash tools/pmc_profile.sh ${TAG}_babyiaxo_xmm_gas_scan32 --workload babyiaxo_xmm_gas_scan32) ;;
    pmc_ascan)  # the fused 16-angle scan workload (2e8-ray launches)
      (cd $ROOT && timeout -k 10 900 bash tools/pmc_profile.sh ${TAG}_babyiaxo_xmm_ascan16 --workload babyiaxo_xmm_ascan16 --rays-per-step 2e8) ;;
    sq)         # sequencer-side counters of the headline kernel (scalar unit, instruction fetch, FIFO stalls)
      (cd $ROOT && timeout -k 10 600 bash tools/pmc_sq.sh ${TAG}_babyiaxo_xmm > gpurun_out/${TAG}_pmc_sq.txt 2>&1) || { tail -20 $ROOT/gpurun_out/${TAG}_pmc_sq.txt; exit 1; }
      tail -3 $ROOT/gpurun_out/${TAG}_pmc_sq.txt ;;
    fullsize)   # BASELINE configs[2] / [1] at full size through the HIP path and the f64 CPU oracle + the throughput table + metric 2
      (cd $ROOT && timeout -k 10 300 python tools/full_size_compare.py > gpurun_out/${TAG}_full_size_compare_1e9.json 2> gpurun_out/${TAG}_fsc.err) || { tail -20 $ROOT/gpurun_out/${TAG}_fsc.err; exit 1; }
      (cd $ROOT && timeout -k 10 200 python tools/full_size_compare.py --workload cast_llnl_gold --rays 1e8 > gpurun_out/${TAG}_full_size_compare_cast_1e8.json 2>> gpurun_out/${TAG}_fsc.err)
      (cd $ROOT && timeout -k 10 200 python tools/full_size_compare.py --workload cast_llnl --rays 1e8 > gpurun_out/${TAG}_full_size_compare_cast_llnl_4coatings_1e8.json 2>> gpurun_out/${TAG}_fsc.err)
      (cd $ROOT && timeout -k 10 200 python tools/full_size_compare.py --workload cast_abrixas --rays 1e8 > gpurun_out/${TAG}_full_size_compare_cast_abrixas_1e8.json 2>> gpurun_out/${TAG}_fsc.err)
      (cd $ROOT && timeout -k 10 200 python tools/full_size_compare.py --workload babyiaxo_xmm_rot --rays 1e8 > gpurun_out/${TAG}_full_size_compare_rot_1e8.json 2>> gpurun_out/${TAG}_fsc.err)
      (cd $ROOT && timeout -k 10 200 python tools/full_size_compare.py --workload babyiaxo_xmm_gas --rays 1e8 > gpurun_out/${TAG}_full_size_compare_gas_1e8.json 2>> gpurun_out/${TAG}_fsc.err)
      (cd $ROOT && timeout -k 10 200 python tools/throughput_table.py > gpurun_out/${TAG}_throughput_table.md 2>&1 && cat gpurun_out/${TAG}_throughput_table.md)
      (cd $ROOT && timeout -k 10 200 python tools/effarea_rms.py > gpurun_out/${TAG}_effective_area_rms.json 2>> gpurun_out/${TAG}_fsc.err)
      tail -c 400 $ROOT/gpurun_out/${TAG}_full_size_compare_1e9.json ;;
    sustained)  # 500-step runs in both accumulation modes, 100 steps of the scan in the integer mode, kernel trace of the scan workload
      (cd $ROOT && timeout -k 10 120 python bench.py --profile-run --steps 500 > gpurun_out/${TAG}_bench_500steps.json 2>/dev/null)
      (cd $ROOT && timeout -k 10 120 python bench.py --profile-run --steps 500 --accumulation fixed64 > gpurun_out/${TAG}_bench_500steps_fixed64.json 2>/dev/null)
      (cd $ROOT && timeout -k 10 120 python bench.py --profile-run --steps 100 --workload babyiaxo_xmm_gas_scan32 --accumulation fixed64 > gpurun_out/${TAG}_bench_scan32_fixed64_100steps.json 2>/dev/null)
      (cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_${TAG}_scan -- python3 $ROOT/bench.py --profile-run --steps 5 --warmup 2 --workload babyiaxo_xmm_gas_scan32 > $ROOT/gpurun_out/prof_${TAG}_scan.log 2>&1) || { tail -20 $ROOT/gpurun_out/prof_${TAG}_scan.log; exit 1; }
      find $ROOT/gpurun_out/prof_${TAG}_scan -name "*kernel_stats.csv" | head -1 | xargs head -4
      python3 -c "
import json
for f in ('bench_500steps', 'bench_500steps_fixed64', 'bench_scan32_fixed64_100steps'):
    d = json.load(open('$ROOT/gpurun_out/${TAG}_' + f + '.json')); print(f, d['value'], d['ms_per_step'], d['results']['flux'], d.get('mass_scan', {}).get('ray_mass_evaluations_per_s'))" ;;
    quick)   # iteration loop: throughput table + instruction-count PMC passes (1-2) for the headline workload and CAST
      (cd $ROOT && timeout -k 10 300 python tools/throughput_table.py > gpurun_out/tt_$TAG.txt 2>&1) || { tail -20 $ROOT/gpurun_out/tt_$TAG.txt; exit 1; }
      cat $ROOT/gpurun_out/tt_$TAG.txt
      (cd $ROOT && PMC_PASSES="1 2" timeout -k 10 600 bash tools/pmc_profile.sh ${TAG}_babyiaxo_xmm --workload babyiaxo_xmm)
      (cd $ROOT && PMC_PASSES="1 2" timeout -k 10 600 bash tools/pmc_profile.sh ${TAG}_cast_llnl_gold --workload cast_llnl_gold --rays-per-step 1e8)
      (cd $ROOT && python tools/pmc_summary.py gpurun_out/pmc_${TAG}_babyiaxo_xmm --rays 1e9 | tail -12 && python tools/pmc_summary.py gpurun_out/pmc_${TAG}_cast_llnl_gold --rays 1e8 | tail -12) ;;
    *) echo "unknown step $STEP"; exit 2 ;;
  esac
  echo "== step $STEP done"
done
