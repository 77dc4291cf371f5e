#!/bin/bash
# Development: the batcher's hint threshold (SYMACCEL_BATCHER_HINT_MB) against caller threads, S = 256 and 64, look-ahead 64
OUT=$PWD/gpurun_out; mkdir -p $OUT; REPO=$PWD
for MB in 1 4 16 32 64 256; do
  for T in 4 16; do
    for S in 64 256; do
      echo -n "hint_mb=$MB T=$T S=$S "; SYMACCEL_BATCHER_HINT_MB=$MB $REPO/symphonia_amd/build/decoders_bench --codec aac --streams $S --lookahead 64 --packets 512 --threads $T | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['packets_per_s']), d['launches'], d['kernel_launches'])"
    done
  done
done 2>&1 | tee $OUT/r05g_hint_sweep.txt
