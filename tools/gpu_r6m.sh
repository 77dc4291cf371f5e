#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_mp3_stereo.py tests/test_gpu_edges.py -m gpu -x -q -k "flac or mp3" 2>&1 | grep -E "passed|failed|error" | tail -3
for w in mp3q flac alac; do
timeout 300 python bench.py --workload $w --no-others --no-cpu-baseline --no-copy-ceiling --no-host-path --repeats 3 --steps 64 2> $OUT/r06m.err > $OUT/r06m_bench_$w.json
python - $OUT/r06m_bench_$w.json $w <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "ms", round(d["ms_per_step"],4), "median", round(d["repeats"]["ms_per_step_median"],4), "frac", round(d["roofline"]["frac"],4), "verified", (d.get("verified") or {}).get("mismatches"))
PY
done
