#!/bin/bash
# Round 6: where the big-block Vorbis pairs lose their time: all-long, all-short and the mixed flag sequence per pair
OUT=$PWD/gpurun_out; mkdir -p $OUT
PAIRS_MIX="0.9,0.7;1,0;0,1;0.9,0.0;0.97,0.7" timeout 600 python tools/vorbis_pairs_probe.py 8,11 9,12 10,13 12,13 7,10 2>&1 | tee $OUT/r06q_vorbis_pairs_mix.txt
