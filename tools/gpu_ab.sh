#!/bin/bash
# A/B a build-time knob on the GPU box: bash tools/gpu_ab.sh <workload> <segment> VAR=val [VAR=val ...]
W=$1; SEG=$2; shift 2
for kv in "$@"; do
  env $kv python -m symphonia_amd.build --force > /dev/null 2>&1
  timeout 120 python bench.py --workload $W --steps 10 --warmup 2 --no-cpu-baseline --segment $SEG 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W $kv seg', d['config']['segment'], 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],4))" | tee -a gpurun_out/sweep.log
done
python -m symphonia_amd.build --force > /dev/null 2>&1
