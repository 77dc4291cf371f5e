#!/bin/bash
# Round 6: the whole GPU suite on the reworked library + the trait-level harness incl. FLAC
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error" | tail -8 | tee $OUT/r06j_gputest.log
B=symphonia_amd/build/decoders_bench
: > $OUT/r06j_decoders.jsonl
timeout 300 $B --codec flac --streams 256 --lookahead 64 --packets 1024 --threads 16 | tee -a $OUT/r06j_decoders.jsonl
timeout 300 $B --codec flac --streams 64 --lookahead 64 --packets 1024 --threads 16 | tee -a $OUT/r06j_decoders.jsonl
timeout 300 $B --codec aac --streams 256 --lookahead 256 --packets 4096 --threads 16 --direct | tee -a $OUT/r06j_decoders.jsonl
