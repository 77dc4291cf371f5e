#!/bin/bash
# Round 6 (last session): the fused MP3 front by instruction count (packed |sample|, SDWA addresses, v_permlane32_swap mid/side): A/B against the round-5 front
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python -m pytest tests/test_mp3_stereo.py tests/test_mp3_packets.py tests/test_mp3_requantize.py tests/test_batcher_kinds.py tests/test_batcher.py -m gpu -q 2>&1 | tail -n 3
rm -f $OUT/r06zz7_ab.log
STEPS=400 WARMUP=50 bash tools/gpu_ab_libs.sh r06zz7 mp3q 3 symphonia_amd/libsymaccel.so build_ab/mp3_front0.so
