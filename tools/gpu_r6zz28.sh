#!/bin/bash
# Round 6 (last session): FLAC config 5 with the first round's wavefronts started in 16 / 64 phases (HBM channel aliasing of rows 16 KiB apart)
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
rm -f $OUT/r06zz28_ab.log
STEPS=20 WARMUP=4 bash tools/gpu_ab_libs.sh r06zz28 flac 2 symphonia_amd/libsymaccel.so build_ab/flac_st16.so build_ab/flac_st64.so
