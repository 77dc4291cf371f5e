#!/bin/bash
# Round 6 (last session): the ALAC sign-LMS update carried as -|res| (A/B against the round-5 form and the v_sad_u32 form) + the slices-on-streams probe
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python -m pytest tests/test_alac.py -m gpu -q 2>&1 | tail -n 3
for v in 0 2; do SYMACCEL_LIB=$PWD/build_ab/alac_u$v.so python -m pytest tests/test_alac.py -m gpu -q 2>&1 | tail -n 1; done
rm -f $OUT/r06zz3_ab.log
STEPS=60 WARMUP=10 bash tools/gpu_ab_libs.sh r06zz3 alac 2 symphonia_amd/libsymaccel.so build_ab/alac_u0.so build_ab/alac_u2.so
for w in vorbisf aactns aacjs; do timeout 300 python tools/overlap_probe.py --workload $w --parts 1 2 4 2>&1 | grep -v -E "^(RCCL|HIP|ROCm)" | tee -a $OUT/r06zz3_overlap.jsonl; done
