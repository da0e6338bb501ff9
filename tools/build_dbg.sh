#!/bin/bash
# Experiment build of libsart with the SART_DEBUG_* knobs (never shipped): tools/microbench/libsart_dbg.so; use with
# SART_LIBSART=$PWD/tools/microbench/libsart_dbg.so.  Extra compiler flags: $1.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
B=/tmp/sart_dbg_build; mkdir -p $B
cd ${SRC:-$ROOT/solaraxionraytracing_amd/csrc}   # SRC: a scratch copy of the sources (tools/exp_rare_counts.sh)
for f in sart_api sart_kernels sart_emission sart_tables sart_opacity; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wall -Wno-unused-function -Wno-bitwise-instead-of-logical -DSART_DEBUG_KNOBS $1 -mllvm -disable-machine-licm -c -o $B/$f.o $f.hip &
done
wait
echo 'extern "C" __attribute__((visibility("default"))) const char* sart_build_id(void) { return "debug-knobs-build"; }' > $B/build_id.cpp
g++ -O1 -fPIC -c -o $B/build_id.o $B/build_id.cpp
/opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -shared -o ${OUT:-$ROOT/tools/microbench/libsart_dbg.so} $B/sart_api.o $B/sart_kernels.o $B/sart_emission.o $B/sart_tables.o $B/sart_opacity.o $B/build_id.o
