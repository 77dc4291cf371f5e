#!/bin/bash
# Round 6: the trait-level harness after the batcher rework (lanes, enqueue outside the mutex, pooled device blocks, completion flags in
# page-locked memory, zero-copy parse, streams out of phase): the sweep INTEGRATION section 7 quotes
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
B=symphonia_amd/build/decoders_bench
: > $OUT/r06e_decoders.jsonl
run() { timeout 300 $B "$@" | tee -a $OUT/r06e_decoders.jsonl; }
for rep in 1 2 3; do
run --codec aac --streams 256 --lookahead 256 --packets 4096 --threads 16 --direct
run --codec aac --streams 256 --lookahead 256 --packets 4096 --threads 16
run --codec aac --streams 256 --lookahead 256 --packets 4096 --threads 16 --direct --in-phase
run --codec mp3h --streams 256 --lookahead 256 --packets 4096 --threads 16 --direct
done
for S in 1 4 16 64 1024; do
run --codec aac --streams $S --lookahead 256 --packets 4096 --threads 16 --direct
done
for S in 1 16 64 256 1024; do
run --codec aac --streams $S --lookahead 64 --packets 2048 --threads 16 --direct
done
run --codec aacd --streams 256 --lookahead 256 --packets 4096 --threads 16
run --codec mp3 --streams 256 --lookahead 256 --packets 4096 --threads 16 --direct
run --codec vorbis --streams 64 --lookahead 64 --packets 1024 --threads 16
run --codec aac --streams 256 --lookahead 256 --packets 2048 --threads 16 --per-stream
