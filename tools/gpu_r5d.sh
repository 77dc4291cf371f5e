#!/bin/bash
# Round 5, fourth GPU call: the full GPU suite, the default line (host_to_host_mp3, decoders, window-major copy probe)
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD
timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 2>&1 | grep -v -E "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -n 12 > $OUT/r05d_gputest.log
cat $OUT/r05d_gputest.log
timeout 900 python bench.py > $OUT/r05d_default_bench.json 2> $OUT/r05d_default_bench.err
echo "default bench rc=$?"; cut -c1-300 $OUT/r05d_default_bench.json; tail -3 $OUT/r05d_default_bench.err
