#!/bin/bash
# Round 6: the trait-level harness with zero-copy parse (--direct) and streams out of phase (default) against in phase
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
B=symphonia_amd/build/decoders_bench
: > $OUT/r06c_decoders.jsonl
run() { timeout 300 $B "$@" | tee -a $OUT/r06c_decoders.jsonl; }
for rep in 1 2; do
run --codec aac --streams 256 --lookahead 256 --packets 4096 --threads 16 --in-phase
run --codec aac --streams 256 --lookahead 256 --packets 4096 --threads 16
run --codec aac --streams 256 --lookahead 256 --packets 4096 --threads 16 --direct --in-phase
run --codec aac --streams 256 --lookahead 256 --packets 4096 --threads 16 --direct
run --codec mp3h --streams 256 --lookahead 256 --packets 4096 --threads 16 --direct
run --codec mp3h --streams 256 --lookahead 256 --packets 4096 --threads 16
done
run --codec aac --streams 256 --lookahead 256 --packets 4096 --threads 8 --direct
run --codec aac --streams 256 --lookahead 256 --packets 4096 --threads 32 --direct
run --codec aac --streams 256 --lookahead 64 --packets 1024 --threads 16 --direct
run --codec aac --streams 1024 --lookahead 64 --packets 1024 --threads 16 --direct
run --codec aac --streams 64 --lookahead 256 --packets 4096 --threads 16 --direct
run --codec aac --streams 16 --lookahead 256 --packets 4096 --threads 16 --direct
run --codec mp3 --streams 256 --lookahead 256 --packets 4096 --threads 16 --direct
