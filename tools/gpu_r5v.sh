#!/bin/bash
# Round 5: FLAC tile store with the pairwise, branch-free decorrelation -- GPU parity, then the config-5 line (twice)
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "flac or Flac" 2>&1 | grep -E "passed|failed|error" | tail -3
for i in 1 2; do
timeout 300 python bench.py --workload flac --no-cpu-baseline --no-copy-ceiling --no-host-path --repeats 3 2> $OUT/r05v.err > $OUT/r05v_bench_flac_$i.json
python - $OUT/r05v_bench_flac_$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("flac ms", round(d["ms_per_step"],4), "median", round(d["repeats"]["ms_per_step_median"],4), "frac", round(d["roofline"]["frac"],4), "verified", (d.get("verified") or {}).get("mismatches"))
PY
done
