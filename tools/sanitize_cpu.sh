#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the CPU-side native code (GPU sanitizers are not available on the pool):
#   libsart_host.so (raytracer_host.cpp, opcd_host.cpp: setup builders, CDFs, H5 / OPCD / CSV readers and writers)
#   libsart_oracle.so (the C restatement the parity tests compare with)
# The instrumented builds temporarily take the place of the shipped ones while the CPU tests that drive them run.
# Usage (repo root, no GPU needed):  bash tools/sanitize_cpu.sh
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
ASAN=$(gcc -print-file-name=libasan.so)
SAN="-O1 -g -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer"
TMP=$(mktemp -d)
restore() {
  [ -f $TMP/libsart_host.so ] && cp $TMP/libsart_host.so $ROOT/solaraxionraytracing_amd/libsart_host.so
  [ -f $TMP/libsart_oracle.so ] && cp $TMP/libsart_oracle.so $ROOT/oracle/libsart_oracle.so
  rm -rf $TMP
}
trap restore EXIT
make -s -C $ROOT/solaraxionraytracing_amd/csrc && make -s -C $ROOT/oracle
cp $ROOT/solaraxionraytracing_amd/libsart_host.so $ROOT/oracle/libsart_oracle.so $TMP/
(cd $ROOT/solaraxionraytracing_amd/csrc && g++ $SAN -std=c++17 -Wall -Wextra -shared -o ../libsart_host.so raytracer_host.cpp opcd_host.cpp -L.. -lsart -ldl -lpthread -Wl,-rpath,'$ORIGIN')
(cd $ROOT/oracle && gcc $SAN -std=gnu11 -fopenmp -ffp-contract=off -fno-fast-math -shared -o libsart_oracle.so sart_oracle.c -lm)
cd $ROOT
ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 LD_PRELOAD=$ASAN OMP_NUM_THREADS=4 \
  python -m pytest tests -q -m "not gpu" -p no:cacheprovider --deselect tests/test_distributed_cpu.py
echo "sanitize_cpu: clean"
