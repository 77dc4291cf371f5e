#!/bin/bash
# Round 6 (last session): ALAC hot loop unrolled 2 / 4 groups of four samples (the history registers' rotation copies) against 1
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for u in 2 4; do SYMACCEL_LIB=$PWD/build_ab/alac_un$u.so python -m pytest tests/test_alac.py -m gpu -q 2>&1 | tail -n 1; done
rm -f $OUT/r06zz15_ab.log
STEPS=60 WARMUP=10 bash tools/gpu_ab_libs.sh r06zz15 alac 2 symphonia_amd/libsymaccel.so build_ab/alac_un2.so build_ab/alac_un4.so
