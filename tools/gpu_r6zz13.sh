#!/bin/bash
# Round 6 (last session): FLAC with G consecutive samples' FMA chains in flight (SYM_FLAC_GROUP 4 = product, 2, 1 = the single chain)
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_flac_packets.py tests/test_packet_fixtures.py tests/test_gpu_fuzz.py -m gpu -q -k "flac or Flac" 2>&1 | tail -n 2
SYMACCEL_LIB=$PWD/build_ab/flac_g2.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py -m gpu -q -k "flac or Flac" 2>&1 | tail -n 1
rm -f $OUT/r06zz13_ab.log
STEPS=20 WARMUP=4 bash tools/gpu_ab_libs.sh r06zz13 flac 2 symphonia_amd/libsymaccel.so build_ab/flac_g2.so build_ab/flac_g1.so
