#!/bin/bash
# Development tool (GPU box): ms per step as a function of how long the timed window is (the board's power management has time
# constants of milliseconds to seconds).   bash tools/steps_sweep.sh <workload> ...
export TMPDIR=/tmp
for W in "$@"; do
  for ws in "3 20" "5 40" "50 200" "200 1000" "1000 4000" "3000 12000" "3 20"; do
    set -- $ws
    python bench.py --workload $W --warmup $1 --steps $2 --no-cpu-baseline --no-others --no-host-path --no-copy-ceiling 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W warmup $1 steps $2: ms_per_step %.4f kernel_ms %.4f frac %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))"
  done
done
