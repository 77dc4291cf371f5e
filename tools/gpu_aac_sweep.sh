#!/bin/bash
# Development tool (GPU box): AAC config 2 at sustained clocks, libraries x segment lengths, alternating.
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
TAG=$1; shift
for rep in 1 2; do
  for spec in "$@"; do
    lib=${spec%%:*}; seg=${spec##*:}; [ "$seg" = "$spec" ] && seg=0
    SYMACCEL_LIB=$lib timeout 120 python bench.py --workload aac --steps 400 --warmup 100 --segment $seg --no-cpu-baseline --no-others --no-host-path --no-copy-ceiling 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('aac', '$(basename $lib)', 'segment $seg', 'ms_per_step', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],4), 'cold', round(d['cold_start']['ms_per_step'],4))" | tee -a $OUT/${TAG}.log
  done
done
