#!/bin/bash
# Round 5, fifth GPU call: the v_dot2_i32_i16 carrier of the FLAC predictor -- GPU parity, config 5's line, kernel stats + SQ counters
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD
timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 -k "flac or Flac or FLAC" 2>&1 | grep -v -E "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -n 8 > $OUT/r05e_gputest_flac.log
cat $OUT/r05e_gputest_flac.log
timeout 600 python bench.py --workload flac --no-others > $OUT/r05e_bench_flac.json 2> $OUT/r05e_bench_flac.err
echo "flac bench rc=$?"; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05e_bench_flac.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("verified"), d.get("repeats",{}).get("ms_per_step"))
PY
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_r05e_flac -o flac -- python $REPO/bench.py --workload flac --steps 10 --warmup 2 --no-cpu-baseline --no-host-path --no-others --no-copy-ceiling > $OUT/prof_r05e_flac.log 2>&1
python $REPO/tools/rocpd_summary.py $(find $OUT/prof_r05e_flac -name '*.db') > $OUT/r05e_flac_rocprofv3.txt 2>&1
rm -rf $OUT/prof_r05e_flac
head -12 $OUT/r05e_flac_rocprofv3.txt | cut -c1-200
cd $REPO
bash tools/gpu_pmc.sh r05e flac
grep -v "narrow\|status\|exp2" $OUT/r05e_flac_sq_counters.txt | cut -c1-60,88-190 | head -30
