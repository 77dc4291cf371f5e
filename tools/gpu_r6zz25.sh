#!/bin/bash
# Round 6 (last session): does the six-tap instantiation cost the order-8 bench anything? (product against the build without it)
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
rm -f $OUT/r06zz25_ab.log
STEPS=60 WARMUP=10 bash tools/gpu_ab_libs.sh r06zz25 alac 3 symphonia_amd/libsymaccel.so build_ab/alac_no6.so
