#!/bin/bash
# One gpurun call: rocprofv3 kernel stats + HBM traffic PMC passes for every workload, then the bench lines (with the CPU
# baseline), which pick up the traffic just measured, then the artifacts (tools/collect_round.py) into
# gpurun_out/profiles_<tag>/.   bash tools/gpu_round.sh [tag];  afterwards, locally: cp gpurun_out/profiles_<tag>/* profiles/
TAG=${1:-r02}
REPO=$PWD
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for w in aac mp3 vorbis flac alac mp3q vorbisf aacjs aactns flacp alacp; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_${TAG}_$w -o $w -- python $REPO/bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline --no-host-path --no-others --no-copy-ceiling > $OUT/prof_${TAG}_$w.log 2>&1
  echo "rocprof stats $w rc=$?"
done
for w in aac mp3 vorbis flac alac mp3q vorbisf aacjs aactns flacp alacp; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_${TAG}_${w}_$c -o $w -- python $REPO/bench.py --workload $w --steps 3 --warmup 1 --no-spinup --no-cpu-baseline --no-host-path --no-others --no-copy-ceiling > $OUT/pmc_${TAG}_${w}_$c.log 2>&1
    echo "rocprof pmc $w $c rc=$?"
  done
done
cd $REPO
python -m pytest tests -m gpu -q 2>&1 | grep -v -E "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -n 5 > $OUT/gputest_${TAG}.log
cat $OUT/gputest_${TAG}.log
python tools/collect_round.py $TAG --traffic-only
for w in aac mp3 vorbis flac alac mp3q vorbisf aacjs aactns flacp alacp; do
  timeout 300 python bench.py --workload $w --steps 20 --warmup 3 --no-others > $OUT/bench_$w.json 2> $OUT/bench_$w.err
  echo "bench $w rc=$?"; tail -n 1 $OUT/bench_$w.json | cut -c1-300
done
python tools/collect_round.py $TAG --stage
cp $OUT/gputest_${TAG}.log $OUT/profiles_${TAG}/${TAG}_gputest.log
# the exact default command (what the driver runs): headline + other_workloads + same-run copy ceilings
timeout 600 python bench.py > $OUT/profiles_${TAG}/${TAG}_default_bench.json 2> $OUT/bench_default.err
echo "default bench rc=$?"; cut -c1-600 $OUT/profiles_${TAG}/${TAG}_default_bench.json
rm -rf $OUT/prof_${TAG}_*/ $OUT/pmc_${TAG}_*/   # the rocpd databases: too large to copy back
