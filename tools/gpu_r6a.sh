#!/bin/bash
# Round 6, first GPU call: the new batch kinds / lanes / per-ticket status on the MI355X, then the trait-level harness with the stats of the
# reworked batcher (mutex wait, host time per launch) -- lanes 1 / 2 / 3, AAC and MP3 from int16 samples
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_batcher.py tests/test_batcher_kinds.py tests/test_lookahead.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error" | tail -5 | tee $OUT/r06a_pytest.log
B=symphonia_amd/build/decoders_bench
: > $OUT/r06a_decoders.jsonl
for lanes in 1 2 3; do
  for rep in 1 2; do
    timeout 300 $B --codec aac --streams 256 --lookahead 256 --packets 4096 --threads 16 --lanes $lanes | tee -a $OUT/r06a_decoders.jsonl
  done
done
timeout 300 $B --codec aac --streams 256 --lookahead 64 --packets 1024 --threads 16 --lanes 2 | tee -a $OUT/r06a_decoders.jsonl
timeout 300 $B --codec aac --streams 1024 --lookahead 64 --packets 1024 --threads 16 --lanes 2 | tee -a $OUT/r06a_decoders.jsonl
timeout 300 $B --codec mp3h --streams 256 --lookahead 256 --packets 4096 --threads 16 --lanes 2 | tee -a $OUT/r06a_decoders.jsonl
timeout 300 $B --codec aacd --streams 256 --lookahead 256 --packets 4096 --threads 16 --lanes 2 | tee -a $OUT/r06a_decoders.jsonl
timeout 300 $B --codec vorbis --streams 64 --lookahead 64 --packets 1024 --threads 16 --lanes 2 | tee -a $OUT/r06a_decoders.jsonl
timeout 300 $B --codec aac --streams 256 --lookahead 256 --packets 4096 --threads 16 --lanes 2 --flush-mb 16 | tee -a $OUT/r06a_decoders.jsonl
timeout 300 $B --codec aac --streams 256 --lookahead 256 --packets 4096 --threads 16 --lanes 2 --flush-mb 256 | tee -a $OUT/r06a_decoders.jsonl
nproc; lscpu | grep "Model name"
