#!/bin/bash
# Round 6 (last session): the two halves of the instruction-count front separately (1: packed requantize, 2: permlane mid/side, 3: both) against the product (0)
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for v in 1 2 3; do SYMACCEL_LIB=$PWD/build_ab/mp3_front$v.so python -m pytest tests/test_mp3_stereo.py -m gpu -q 2>&1 | tail -n 1; done
rm -f $OUT/r06zz8_ab.log
STEPS=400 WARMUP=50 bash tools/gpu_ab_libs.sh r06zz8 mp3q 2 symphonia_amd/libsymaccel.so build_ab/mp3_front1.so build_ab/mp3_front2.so build_ab/mp3_front3.so
