#!/bin/bash
# Round 6 (last session): ALAC with a six-tap steady instantiation for wavefronts of orders <= 6 (ffmpeg's 4 .. 6): by order mix, against the build without it
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
SYM_FUZZ_ITERS=30 python -m pytest tests/test_alac.py tests/test_alac_packets.py tests/test_gpu_fuzz.py tests/test_batcher_kinds.py -m gpu -q -k "alac or Alac" 2>&1 | tail -n 1
for lib in symphonia_amd/libsymaccel.so build_ab/alac_no6.so; do echo "## $lib"; SYMACCEL_LIB=$PWD/$lib python tools/alac_orders_time.py 2>&1 | grep orders; done | tee $OUT/r06zz23_alac_orders.txt
