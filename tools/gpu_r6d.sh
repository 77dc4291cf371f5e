#!/bin/bash
# Round 6: copy-kernel variants (non-temporal accesses, capped grids); slot peaks
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
B=symphonia_amd/build/decoders_bench
: > $OUT/r06d_decoders.jsonl
for cfg in "0 256" "1 256" "0 128" "1 128" "0 0" "1 0"; do
  set -- $cfg
  for codec in aac mp3h; do
    for rep in 1 2; do
    echo "{\"nt\": $1, \"wgs\": $2}" >> $OUT/r06d_decoders.jsonl
    SYMACCEL_BATCH_COPY_NT=$1 SYMACCEL_BATCH_COPY_WGS=$2 timeout 300 $B --codec $codec --streams 256 --lookahead 256 --packets 4096 --threads 16 --direct | tee -a $OUT/r06d_decoders.jsonl
    done
  done
done
