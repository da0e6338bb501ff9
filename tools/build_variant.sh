#!/bin/bash
# A second build of libsart.so for same-box A/B runs (tools/exp_ab.py lib1.so lib2.so): the Makefile's flags, sources from
# $SRC (default: the tree's csrc; e.g. a `git worktree` of another commit), output $OUT (default tools/microbench/libsart_B.so).
# Extra compiler flags: $1.  Use with SART_LIBSART=<path>.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=${SRC:-$ROOT/solaraxionraytracing_amd/csrc}
OUT=${OUT:-$ROOT/tools/microbench/libsart_B.so}
B=$(mktemp -d /tmp/sart_variant_XXXX)
cd $SRC
for f in sart_api sart_kernels sart_emission sart_tables sart_opacity; do
  EXTRA=""
  case $f in sart_tables|sart_opacity) EXTRA="-ffp-contract=off";; esac
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Wall -Wno-unused-function -Wno-bitwise-instead-of-logical $EXTRA $1 -mllvm -disable-machine-licm -c -o $B/$f.o $f.hip &
done
wait
echo "extern \"C\" __attribute__((visibility(\"default\"))) const char* sart_build_id(void) { return \"variant-$(basename $OUT .so)\"; }" > $B/build_id.cpp
g++ -O1 -fPIC -c -o $B/build_id.o $B/build_id.cpp
/opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -shared -o $OUT $B/sart_api.o $B/sart_kernels.o $B/sart_emission.o $B/sart_tables.o $B/sart_opacity.o $B/build_id.o
rm -rf $B
echo built $OUT
