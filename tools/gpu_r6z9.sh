#!/bin/bash
# Round 6: the TNS pass with its filters in stream order (what a decoder hands over) against the shuffled list of the bench
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for order in stream shuffled; do for direct in 1 0; do
  ( cd /tmp; SYM_BENCH_TNS_ORDER=$order SYMACCEL_TNS_DIRECT=$direct timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_r06z9 -o aactns -- python $OLDPWD/bench.py --workload aactns --steps 20 --warmup 2 --no-cpu-baseline --no-host-path --no-others --no-copy-ceiling --repeats 0 > $OUT/r06z9_bench_${order}_direct$direct.json 2>/dev/null )
  python tools/rocpd_summary.py gpurun_out/prof_r06z9/aactns_results.db > $OUT/r06z9_aactns_${order}_direct${direct}_rocprofv3.txt 2>&1; echo "== $order direct $direct"; sed -n 3,5p $OUT/r06z9_aactns_${order}_direct${direct}_rocprofv3.txt | cut -c1-200; rm -rf $OUT/prof_r06z9
  python -c "
import json,sys
d=json.loads(open('$OUT/r06z9_bench_${order}_direct$direct.json').read().strip().splitlines()[-1]); print('ms', d['ms_per_step'], 'verified', (d.get('verified') or {}).get('mismatches'))"
done; done
