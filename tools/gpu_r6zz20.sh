#!/bin/bash
# Round 6 (last session): the ILP scheduling strategy for aac.hip (the quad walk, with and without joint stereo on load)
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
rm -f $OUT/r06zz20_ab.log
for w in aac aacjs; do STEPS=400 WARMUP=50 bash tools/gpu_ab_libs.sh r06zz20 $w 2 symphonia_amd/libsymaccel.so build_ab/ilp_aac.so; done
