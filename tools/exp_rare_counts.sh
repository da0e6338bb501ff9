#!/bin/bash
# Experiment build that counts how often each divergent region of the ray kernel runs (profiles/EXPERIMENTS.md Part I §15):
# a scratch copy of csrc/ + tools/exp_rare_counts.patch (a counter per region, bumped by its first active lane), built like
# tools/build_dbg.sh into tools/microbench/libsart_count.so.  Then, on the GPU box:
#   SART_LIBSART=$PWD/tools/microbench/libsart_count.so python tools/exp_rare_counts.py
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
S=/tmp/sart_count_src; rm -rf $S; mkdir -p $S
cp $ROOT/solaraxionraytracing_amd/csrc/*.hip $ROOT/solaraxionraytracing_amd/csrc/*.h $ROOT/solaraxionraytracing_amd/csrc/*.cpp $S/
(cd $S && patch -p1 < $ROOT/tools/exp_rare_counts.patch)
SRC=$S OUT=$ROOT/tools/microbench/libsart_count.so bash $ROOT/tools/build_dbg.sh "-DSART_COUNT_RARE -I$ROOT/solaraxionraytracing_amd/csrc -I$ROOT/include"
