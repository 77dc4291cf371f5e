#!/bin/bash
# Round 5: floor-1 render (byte plane) on the f32 side + v_cvt_pk_u8_f32 -- GPU parity of everything floor-related, then the posts -> PCM line
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "floor or vorbis" 2>&1 | grep -E "passed|failed|error" | tail -3
for i in 1 2; do
timeout 300 python bench.py --workload vorbisf --no-cpu-baseline --no-copy-ceiling --no-host-path --repeats 3 2> $OUT/r05u.err > $OUT/r05u_bench_vorbisf_$i.json
python - $OUT/r05u_bench_vorbisf_$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("vorbisf ms", round(d["ms_per_step"],4), "median", round(d["repeats"]["ms_per_step_median"],4), "frac", round(d["roofline"]["frac"],4), "verified", (d.get("verified") or {}).get("mismatches"))
PY
done
