#!/bin/bash
# Round 6 (last session): FLAC tap order -- oldest sample first (consecutive samples' FMA chains overlap) against newest first
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_flac_packets.py tests/test_packet_fixtures.py -m gpu -q -k "flac or Flac" 2>&1 | tail -n 2
rm -f $OUT/r06zz12_ab.log
STEPS=20 WARMUP=4 bash tools/gpu_ab_libs.sh r06zz12 flac 2 symphonia_amd/libsymaccel.so build_ab/flac_newest.so
