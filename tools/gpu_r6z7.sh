#!/bin/bash
# Round 6: the TNS pass with lane-by-lane movement in a small pass
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_aac_tools.py tests/test_aac_js_fused.py tests/test_gpu_fuzz.py tests/test_aac_packets.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8
show() { python - $1 $2 <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "ms", round(d["ms_per_step"],4), "median", round(d["repeats"]["ms_per_step_median"],4) if d.get("repeats") else None, "frac", round(d["roofline"]["frac"],4), "verified", (d.get("verified") or {}).get("mismatches"))
PY
}
for direct in 1 0; do
SYMACCEL_TNS_DIRECT=$direct timeout 600 python bench.py --workload aactns --no-others --no-cpu-baseline --no-host-path --no-copy-ceiling --repeats 2 > $OUT/r06z7_bench_aactns_direct$direct.json 2> $OUT/r06z7_aactns.err; show $OUT/r06z7_bench_aactns_direct$direct.json aactns_direct$direct
( cd /tmp; SYMACCEL_TNS_DIRECT=$direct timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_r06z7_aactns -o aactns -- python $OLDPWD/bench.py --workload aactns --steps 10 --warmup 2 --no-cpu-baseline --no-host-path --no-others --no-copy-ceiling --repeats 0 > /dev/null 2>&1 )
python tools/rocpd_summary.py gpurun_out/prof_r06z7_aactns/aactns_results.db > $OUT/r06z7_aactns_direct${direct}_rocprofv3.txt 2>&1; head -10 $OUT/r06z7_aactns_direct${direct}_rocprofv3.txt | cut -c1-200; rm -rf $OUT/prof_r06z7_aactns
done
