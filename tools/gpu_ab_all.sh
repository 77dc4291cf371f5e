#!/bin/bash
# A/B one build-time knob over every workload on the GPU box: bash tools/gpu_ab_all.sh VAR=a VAR=b ...
for kv in "$@"; do
  env $kv python -m symphonia_amd.build --force > /dev/null 2>&1
  for W in aac mp3 vorbis flac alac; do
    for rep in 1 2; do
      timeout 120 python bench.py --workload $W --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W $kv', 'ms_per_step', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],4))" | tee -a gpurun_out/sweep.log
    done
  done
done
python -m symphonia_amd.build --force > /dev/null 2>&1
