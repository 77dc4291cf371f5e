#!/bin/bash
# Quick GPU iteration: parity tests + bench sweeps without the CPU baseline.  bash tools/gpu_quick.sh
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x --timeout 180 2>&1 | tail -5 | tee $OUT/pytest_gpu.log
for seg in 16 32 64 128; do
  timeout 120 python bench.py --workload mp3 --steps 10 --warmup 2 --no-cpu-baseline --segment $seg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mp3 seg', d['config']['segment'], 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],4))" | tee -a $OUT/sweep.log
done
for seg in 8 16 32 64; do
  timeout 120 python bench.py --workload aac --steps 10 --warmup 2 --no-cpu-baseline --segment $seg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('aac seg', d['config']['segment'], 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],4))" | tee -a $OUT/sweep.log
done
for w in flac vorbis; do
  timeout 120 python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],4))" | tee -a $OUT/sweep.log
done
