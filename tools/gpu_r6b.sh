#!/bin/bash
# Round 6: where a launch's host time goes (HIP API calls against descriptor building), by caller threads
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
B=symphonia_amd/build/decoders_bench
: > $OUT/r06b_decoders.jsonl
for T in 1 4 8 16 32; do
  timeout 300 $B --codec aac --streams 256 --lookahead 256 --packets 4096 --threads $T --lanes 2 | tee -a $OUT/r06b_decoders.jsonl
done
for fm in 8 32; do
  timeout 300 $B --codec aac --streams 256 --lookahead 256 --packets 4096 --threads 16 --lanes 1 --flush-mb $fm | tee -a $OUT/r06b_decoders.jsonl
done
