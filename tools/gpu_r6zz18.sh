#!/bin/bash
# Round 6 (last session): soak -- the GPU fuzz suite with 60 iterations per test (default 6) on the final library
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
SYM_FUZZ_ITERS=60 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -n 4 | tee $OUT/r06zz18_fuzz_soak.log
