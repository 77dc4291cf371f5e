#!/bin/bash
# Round 6: is the small TNS pass bound by all filters touching the same offset of their 4 KiB frames at the same time?
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for range in same staggered; do
  ( cd /tmp; SYM_BENCH_TNS_RANGE=$range timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_r06z10 -o aactns -- python $OLDPWD/bench.py --workload aactns --steps 20 --warmup 2 --no-cpu-baseline --no-host-path --no-others --no-copy-ceiling --repeats 0 --no-verify > /dev/null 2>&1 )
  python tools/rocpd_summary.py gpurun_out/prof_r06z10/aactns_results.db > $OUT/r06z10_aactns_${range}_rocprofv3.txt 2>&1; echo "== $range"; sed -n 3,5p $OUT/r06z10_aactns_${range}_rocprofv3.txt | cut -c1-200; rm -rf $OUT/prof_r06z10
done
