#!/bin/bash
# Round 6 (last session): MP3 config 3 (f32 spectra in) at TWO wavefronts per SIMD -- how much of the fused kernel's extra time is occupancy, how much the front
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
rm -f $OUT/r06zz9_ab.log
STEPS=400 WARMUP=50 bash tools/gpu_ab_libs.sh r06zz9 mp3 2 symphonia_amd/libsymaccel.so build_ab/mp3_w2.so
