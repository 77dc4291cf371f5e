#!/bin/bash
# Round 6: the small TNS pass by groups in flight (1 .. 4), and without its arithmetic
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_aac_tools.py tests/test_aac_js_fused.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8
for v in product tns_ldsx0 tns_ahead1; do
  L=$PWD/build_ab/$v/libsymaccel.so; [ $v = product ] && L=$PWD/symphonia_amd/libsymaccel.so
  V=""; [ $v != product ] && V="--no-verify"
  ( cd /tmp; SYMACCEL_LIB=$L timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_r06z8 -o aactns -- python $OLDPWD/bench.py --workload aactns --steps 20 --warmup 2 --no-cpu-baseline --no-host-path --no-others --no-copy-ceiling --repeats 0 $V > /dev/null 2>&1 )
  python tools/rocpd_summary.py gpurun_out/prof_r06z8/aactns_results.db > $OUT/r06z8_aactns_${v}_rocprofv3.txt 2>&1; echo "== $v"; sed -n 3,5p $OUT/r06z8_aactns_${v}_rocprofv3.txt | cut -c1-200; rm -rf $OUT/prof_r06z8
done
