#!/bin/bash
# Samples socket power, shader clock and junction temperature (rocm-smi) while the ray kernel runs:
#   bash tools/power_sample.sh [power_run.py args]      (environment: SART_LIBSART / SART_DEBUG_FLAGS for ablated builds)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
python3 $ROOT/tools/power_run.py --seconds ${POWER_SECONDS:-14} "$@" > /tmp/power_run.txt 2>/tmp/power_run.err &
PID=$!
sleep ${POWER_DELAY:-9}
for i in $(seq 1 ${POWER_SAMPLES:-6}); do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Package Power|sclk|junction" | sed 's/.*: //' | tr '\n' ' '
  echo
  sleep 1
done
wait $PID
cat /tmp/power_run.txt; tail -2 /tmp/power_run.err
