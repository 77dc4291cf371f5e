#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
B=symphonia_amd/build/decoders_bench
: > $OUT/r06k_decoders.jsonl
for hint in 4 32 128; do
  SYMACCEL_BATCHER_HINT_MB=$hint timeout 300 $B --codec flac --streams 256 --lookahead 64 --packets 1024 --threads 16 --lanes 2 | tee -a $OUT/r06k_decoders.jsonl
done
SYMACCEL_BATCHER_HINT_MB=128 timeout 300 $B --codec flac --streams 256 --lookahead 64 --packets 1024 --threads 16 --lanes 2 --flush-mb 256 | tee -a $OUT/r06k_decoders.jsonl
timeout 300 $B --codec flac --streams 256 --lookahead 64 --packets 1024 --threads 16 --lanes 2 --in-phase | tee -a $OUT/r06k_decoders.jsonl
timeout 300 $B --codec flac --streams 256 --lookahead 64 --packets 1024 --threads 16 --lanes 2 --in-phase --flush-mb 512 | tee -a $OUT/r06k_decoders.jsonl
