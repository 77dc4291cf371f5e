#!/bin/bash
# Round 6 (last session): the instruction-count front again with the scale address as an LDS integer address (no add per line): timing + SQ counters
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for v in 1 3; do SYMACCEL_LIB=$PWD/build_ab/mp3_front$v.so python -m pytest tests/test_mp3_stereo.py tests/test_mp3_packets.py -m gpu -q 2>&1 | tail -n 1; done
rm -f $OUT/r06zz11_ab.log
STEPS=400 WARMUP=50 bash tools/gpu_ab_libs.sh r06zz11 mp3q 3 symphonia_amd/libsymaccel.so build_ab/mp3_front1.so build_ab/mp3_front3.so
export SYMACCEL_LIB=$PWD/build_ab/mp3_front3.so
bash tools/gpu_pmc.sh r06zz11_front3 mp3q
grep -E "SQ_INSTS|SQ_ACTIVE|SQ_WAIT|SQ_LDS" gpurun_out/r06zz11_front3_mp3q_sq_counters.txt | cut -c1-30,88-150
