#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python bench.py --workload aactns --no-others --no-cpu-baseline --no-host-path --no-copy-ceiling --repeats 2 > $OUT/r06l_bench_aactns.json 2> $OUT/r06l_aactns.err; tail -3 $OUT/r06l_aactns.err
timeout 900 python bench.py > $OUT/r06l_default_bench.json 2> $OUT/r06l_default.err; tail -2 $OUT/r06l_default.err
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_r06l_aactns -o aactns -- python $OLDPWD/bench.py --workload aactns --steps 10 --warmup 2 --no-cpu-baseline --no-host-path --no-others --no-copy-ceiling --repeats 0 > /dev/null 2>&1
cd $OLDPWD; python tools/rocpd_summary.py gpurun_out/prof_r06l_aactns/aactns_results.db > $OUT/r06l_aactns_rocprofv3.txt 2>&1; head -12 $OUT/r06l_aactns_rocprofv3.txt; rm -rf $OUT/prof_r06l_aactns
