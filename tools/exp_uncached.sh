#!/bin/bash
# Experiment: hot gather tables in MTYPE_UC memory (hipDeviceMallocUncached: reads bypass L2, 32/64-byte requests at the fabric
# instead of 128-byte lines).  One library (tools/microbench/libsart_U.so, built from a scratch copy with the SART_UNCACHED_TABLES
# knob), alternating settings of the knob.
export SART_AB_RAYS=${SART_AB_RAYS:-1e9} SART_AB_REPS=${SART_AB_REPS:-5} SART_AB_WORKLOADS=${SART_AB_WORKLOADS:-BabyIAXO,CAST}
for round in 1 2; do
  for t in none refl cdf guide etab refl,cdf; do
    echo "== uncached: $t"
    SART_UNCACHED_TABLES=$t python tools/exp_ab.py tools/microbench/libsart_U.so | grep -v libsart
  done
done
