#!/bin/bash
# Round 6 (last session): floor-1 byte render with sixteen consecutive lines per lane (one prefix maximum, one 16-byte map read and store per 16 lines) against four groups of four
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python -m pytest tests/test_vorbis_floor_y.py tests/test_vorbis_decode.py tests/test_vorbis_packets.py tests/test_batcher_kinds.py tests/test_gpu_parity.py -m gpu -q -k "vorbis or Vorbis or floor" 2>&1 | tail -n 2
rm -f $OUT/r06zz17_ab.log
STEPS=400 WARMUP=50 bash tools/gpu_ab_libs.sh r06zz17 vorbisf 3 symphonia_amd/libsymaccel.so build_ab/f1_lane0.so
