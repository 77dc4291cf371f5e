#!/bin/bash
# Round 6: one launch for the TNS tap classes, no chain index when every chain is paired: aactns and aacjs again (+ the AAC GPU tests)
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_aac_tools.py tests/test_aac_js_fused.py tests/test_gpu_fuzz.py tests/test_aac_packets.py tests/test_gpu_parity.py tests/test_batcher_kinds.py -m gpu -x -q -k "aac or tns or AAC" 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8
show() { python - $1 $2 <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "ms", round(d["ms_per_step"],4), "median", round(d["repeats"]["ms_per_step_median"],4) if d.get("repeats") else None, "frac", round(d["roofline"]["frac"],4), "verified", (d.get("verified") or {}).get("mismatches"))
PY
}
for w in aactns aacjs; do
timeout 600 python bench.py --workload $w --no-others --no-cpu-baseline --no-host-path --no-copy-ceiling --repeats 2 > $OUT/r06z12_bench_$w.json 2> $OUT/r06z12_$w.err; show $OUT/r06z12_bench_$w.json $w
( cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_r06z12 -o $w -- python $OLDPWD/bench.py --workload $w --steps 20 --warmup 2 --no-cpu-baseline --no-host-path --no-others --no-copy-ceiling --repeats 0 > /dev/null 2>&1 )
python tools/rocpd_summary.py gpurun_out/prof_r06z12/${w}_results.db > $OUT/r06z12_${w}_rocprofv3.txt 2>&1; head -9 $OUT/r06z12_${w}_rocprofv3.txt | cut -c1-200; rm -rf $OUT/prof_r06z12
done
