#!/bin/bash
# Round 6 (last session): ALAC -- the direction mask seeded by one signum, the prediction sum as one chain (oldest first): against the previous build
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python -m pytest tests/test_alac.py tests/test_alac_packets.py -m gpu -q 2>&1 | tail -n 1
rm -f $OUT/r06zz14_ab.log
STEPS=60 WARMUP=10 bash tools/gpu_ab_libs.sh r06zz14 alac 3 symphonia_amd/libsymaccel.so build_ab/alac_prev.so
