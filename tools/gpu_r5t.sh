#!/bin/bash
# Round 5: the FLAC integer-sum kernel with vcc carry-outs -- GPU parity (wide coefficients take it), then the whole GPU suite
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "flac" 2>&1 | tail -3
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
