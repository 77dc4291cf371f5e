#!/bin/bash
# Round 6 (last session): the small TNS pass with ONE filter per lane on twice the wavefronts (SYM_TNS_SPLIT) -- is its movement bound by requests in flight?
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
SYMACCEL_LIB=$PWD/build_ab/tns_split.so python -m pytest tests/test_aac_tools.py tests/test_aac_js_fused.py -m gpu -q 2>&1 | tail -n 1
rm -f $OUT/r06zz16_ab.log
STEPS=400 WARMUP=50 bash tools/gpu_ab_libs.sh r06zz16 aactns 3 symphonia_amd/libsymaccel.so build_ab/tns_split.so
