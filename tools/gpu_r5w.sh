#!/bin/bash
# Round 5: the wavefront index through v_readfirstlane in the walk kernels -- GPU parity of the whole suite, then the lines it touches
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
for w in aac vorbis mp3q aacjs; do
timeout 300 python bench.py --workload $w --no-others --no-cpu-baseline --no-copy-ceiling --no-host-path --repeats 3 2> $OUT/r05w.err > $OUT/r05w_bench_$w.json
python - $OUT/r05w_bench_$w.json $w <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "ms", round(d["ms_per_step"],4), "median", round(d["repeats"]["ms_per_step_median"],4), "frac", round(d["roofline"]["frac"],4))
PY
done
