#!/bin/bash
# Alternating A/B of prebuilt libraries on the GPU box: bash tools/gpu_ab_libs.sh <tag> <workload> <reps> lib1.so lib2.so ...
# (each line: workload, library, ms per step, launch period, roofline fraction; appended to gpurun_out/<tag>_ab.log)
TAG=$1; W=$2; REPS=$3; shift 3
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for rep in $(seq $REPS); do
  for lib in "$@"; do
    SYMACCEL_LIB=$lib timeout 120 python bench.py --workload $W --steps ${STEPS:-400} --warmup ${WARMUP:-100} --no-cpu-baseline --no-others --no-host-path --no-copy-ceiling 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W', '$(basename $lib)', 'ms_per_step', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],4))" | tee -a $OUT/${TAG}_ab.log
  done
done
