#!/bin/bash
# Round 5: Vorbis 256/2048 walk -- three workgroups per CU (LDS 49.5 KiB, 116 VGPRs allow it) by segment length
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for w in vorbis vorbisf; do
for seg in 0 128 86 100 64 43; do
  timeout 200 python bench.py --workload $w --segment $seg --no-others --no-cpu-baseline --no-host-path --no-copy-ceiling --repeats 3 2> $OUT/r05s.err > $OUT/r05s_${w}_seg$seg.json
  python - $OUT/r05s_${w}_seg$seg.json $w $seg <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "segment", sys.argv[3], "ms", round(d["ms_per_step"],4), "median", round(d["repeats"]["ms_per_step_median"],4), "frac", round(d["roofline"]["frac"],4))
PY
done
done
