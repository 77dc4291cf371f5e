#!/bin/bash
# Round 5: a soak of the trait-level path -- every codec of the harness, many packets, 16 caller threads, failures must stay 0
OUT=$PWD/gpurun_out; mkdir -p $OUT
B=symphonia_amd/build/decoders_bench
for spec in "aac 256 64 4096" "aacd 256 256 2048" "mp3 256 64 4096" "mp3h 256 256 4096" "vorbis 64 64 2048" "aac 1024 64 1024" "aacd 64 16 2048" "mp3h 7 33 3001" "aac 3 5 2000"; do
  set -- $spec
  timeout 300 $B --codec $1 --streams $2 --lookahead $3 --packets $4 --threads 16 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$spec', '->', 'packets', d['packets'], 'failures', d['failures'], 'packets_per_s', round(d['packets_per_s']), 'launches', d['launches'], 'max_chains_per_launch', d['max_chains_per_launch'])"
done
