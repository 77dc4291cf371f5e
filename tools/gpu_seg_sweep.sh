#!/bin/bash
# Development tool (GPU box): a workload at sustained clocks over --segment values (units per walk; 0 = the library's choice).
#   bash tools/gpu_seg_sweep.sh <tag> <workload> seg seg ...
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
TAG=$1; W=$2; shift 2
for rep in 1 2; do
  for seg in "$@"; do
    timeout 120 python bench.py --workload $W --steps 400 --warmup 100 --segment $seg --no-cpu-baseline --no-others --no-host-path --no-copy-ceiling 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W', 'segment $seg', 'ms_per_step', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],4))" | tee -a $OUT/${TAG}.log
  done
done
