#!/bin/bash
# Round 6 (last session): ALAC small-order instantiations at 2 / 3 / 4 wavefronts per SIMD with the new update
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for v in w4 w2; do SYMACCEL_LIB=$PWD/build_ab/alac_$v.so python -m pytest tests/test_alac.py -m gpu -q 2>&1 | tail -n 1; done
rm -f $OUT/r06zz4_ab.log
STEPS=60 WARMUP=10 bash tools/gpu_ab_libs.sh r06zz4 alac 2 symphonia_amd/libsymaccel.so build_ab/alac_w4.so build_ab/alac_w2.so
