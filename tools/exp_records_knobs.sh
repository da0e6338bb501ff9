#!/bin/bash
# sart_trace_records into a fresh 2e7-record buffer for 2 / 4 / 8 / 16 pre-fault threads and two chunk sizes
# (SART_PREFAULT_THREADS, SART_RECORDS_CHUNK): profiles/r03_exp_records_knobs.txt.  Run on the GPU box.
for T in 2 4 8 16; do
  for CH in 1048576 524288; do
    echo "threads $T chunk $CH"
    SART_PREFAULT_THREADS=$T SART_RECORDS_CHUNK=$CH timeout -k 10 200 python tools/records_rate.py --out gpurun_out/rr_tmp.json \
      | grep -A2 '"fresh_buffer": {' | grep records_per_s
  done
done
