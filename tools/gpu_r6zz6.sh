#!/bin/bash
# Round 6 (last session): ALAC with the carried residual in the wide form too ("mid" wavefronts)
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python -m pytest tests/test_alac.py tests/test_alac_packets.py -m gpu -q 2>&1 | tail -n 1
rm -f $OUT/r06zz6_ab.log
STEPS=60 WARMUP=10 bash tools/gpu_ab_libs.sh r06zz6 alac 2 symphonia_amd/libsymaccel.so
python tools/alac_widths_time.py 2>&1 | grep -v -E "^(RCCL|HIP|ROCm|/opt)" | tee $OUT/r06zz6_alac_widths.jsonl
