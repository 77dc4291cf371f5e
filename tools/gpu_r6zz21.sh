#!/bin/bash
# Round 6 (last session): floor-1 byte render, blocks of <= 256 lines with four lines per lane (32 lanes busy on a 128-line block) against sixteen for every size (SYM_F1_LANE16 = 2)
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
SYM_FUZZ_ITERS=30 python -m pytest tests/test_vorbis_floor_y.py tests/test_vorbis_decode.py tests/test_vorbis_packets.py tests/test_batcher_kinds.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -k "vorbis or Vorbis or floor" 2>&1 | tail -n 2
rm -f $OUT/r06zz21_ab.log
STEPS=400 WARMUP=50 bash tools/gpu_ab_libs.sh r06zz21 vorbisf 3 symphonia_amd/libsymaccel.so build_ab/f1_lane2.so
