#!/bin/bash
# Round 5: the AAC walk over a window-major layout (tuned builds, timing only): pad / offset variants beside the product, list walked twice
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for round in 1 2; do
  for v in product wm_0_0 wm_1_0 wm_0_1 wm_3_0 wm_7_1; do
    if [ $v = product ]; then unset SYMACCEL_LIB; else export SYMACCEL_LIB=$PWD/symphonia_amd/build/tuned/$v/libsymaccel.so; fi
    timeout 200 python tools/aac_offsets.py --few 2> $OUT/r05o_$v.err | tee -a $OUT/r05o_aac_window_major.jsonl
    echo
  done
done
