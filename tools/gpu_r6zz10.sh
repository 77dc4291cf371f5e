#!/bin/bash
# Round 6 (last session): SQ counters of the fused MP3 kernel with the instruction-count front (variant 3) -- does the dynamic instruction count drop?
export SYMACCEL_LIB=$PWD/build_ab/mp3_front3.so
bash tools/gpu_pmc.sh r06zz10_front3 mp3q
grep -E "SQ_INSTS|SQ_ACTIVE|SQ_WAIT|SQ_LDS|avg_us" gpurun_out/r06zz10_front3_mp3q_sq_counters.txt | cut -c1-30,88-190
