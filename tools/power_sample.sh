#!/bin/bash
# Samples socket power and shader clock (rocm-smi) while the headline workload runs: bash tools/power_sample.sh [bench args]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
python3 $ROOT/bench.py --profile-run --steps 500 --warmup 5 "$@" > /tmp/power_bench.json 2>/tmp/power_bench.err &
PID=$!
sleep 12
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (junction|edge)" | tr -s ' ' | tr '\n' ';'
  echo
  sleep 0.7
done
wait $PID
python3 -c "import json; d=json.load(open('/tmp/power_bench.json')); print('rays/s %.4g  ms_per_step %.3f' % (d['value'], d['ms_per_step']))"
