#!/bin/bash
# Round 6: the link's two directions in the trait-level pipeline: kernel trace of decoders_bench with the copy kernels tagged by direction
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD
B=$REPO/symphonia_amd/build/decoders_bench
export LD_LIBRARY_PATH=$REPO/symphonia_amd:$LD_LIBRARY_PATH
cd /tmp
: > $OUT/r06t_copy_timeline.txt
for cfg in "2 256" "1 256" "2 128" "1 1024"; do
  set -- $cfg
  echo "== lanes $1, copy grid cap $2: decoders_bench --codec aac --streams 256 --lookahead 256 --packets 2048 --threads 16 --direct --lanes $1" | tee -a $OUT/r06t_copy_timeline.txt
  SYMACCEL_BATCH_COPY_WGS=$2 timeout 300 rocprofv3 --kernel-trace -d $OUT/kt_r06t -o p -- $B --codec aac --streams 256 --lookahead 256 --packets 2048 --threads 16 --direct --lanes $1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('packets/s under the profiler', round(d['packets_per_s']), 'launches', d['launches'], 'GB/s each way', d['GBps_each_way'])" | tee -a $OUT/r06t_copy_timeline.txt
  python $REPO/tools/copy_timeline.py $(find $OUT/kt_r06t -name '*.db' | head -1) 0.4 | tee -a $OUT/r06t_copy_timeline.txt
  rm -rf $OUT/kt_r06t
done
