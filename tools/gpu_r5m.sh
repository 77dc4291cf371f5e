#!/bin/bash
# Round 5: the stride question -- copy probes around the walk's 1 MiB workgroup spacing, then the AAC headline at segment lengths off the power of two
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python tools/probe_layout.py > $OUT/r05m_probe_layout.json 2> $OUT/r05m_probe_layout.err; echo "probe rc=$?"
cat $OUT/r05m_probe_layout.json | cut -c1-3000
for seg in 0 256 260 264 272 288 320 128 132; do
  timeout 200 python bench.py --segment $seg --no-others --no-cpu-baseline --no-host-path --no-copy-ceiling --repeats 3 > $OUT/r05m_aac_seg$seg.json 2> $OUT/r05m_aac_seg$seg.err
  python - $OUT/r05m_aac_seg$seg.json $seg <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("segment", sys.argv[2], "ms", round(d["ms_per_step"],4), "frac", round(d["roofline"]["frac"],4), "median", d.get("repeats",{}).get("value_median") if d.get("repeats") else None)
PY
done
