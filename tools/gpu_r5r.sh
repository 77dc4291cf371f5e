#!/bin/bash
# Round 5: MP3 walk, segment length against the unequal progress of a SIMD's three wavefronts (more, shorter segments = the dispatcher balances)
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for w in mp3 mp3q; do
for seg in 0 64 43 32 22 16 11; do
  timeout 200 python bench.py --workload $w --segment $seg --no-others --no-cpu-baseline --no-host-path --no-copy-ceiling --repeats 3 2> $OUT/r05r.err > $OUT/r05r_${w}_seg$seg.json
  python - $OUT/r05r_${w}_seg$seg.json $w $seg <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "segment", sys.argv[3], "ms", round(d["ms_per_step"],4), "median", round(d["repeats"]["ms_per_step_median"],4), "frac", round(d["roofline"]["frac"],4))
PY
done
done
