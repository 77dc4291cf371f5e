#!/bin/bash
# Quick GPU iteration: parity tests + bench sweeps without the CPU baseline.  bash tools/gpu_quick.sh [workloads...]
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x --timeout 180 2>&1 | tail -5 | tee $OUT/pytest_gpu.log
one() {  # workload segment
  timeout 120 python bench.py --workload $1 --steps 10 --warmup 2 --no-cpu-baseline --segment $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 seg', d['config']['segment'], 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],4))" | tee -a $OUT/sweep.log
}
for w in ${@:-aac mp3 vorbis flac}; do
  one $w 0
  if [ $w != flac ]; then for seg in 16 32 64; do one $w $seg; done; fi
done
