#!/bin/bash
# Round 5: kernel stats of the trait-level harness (gather / synthesis / scatter kernels of the batcher) + the N > 1 C path on one GPU
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD
cd /tmp
for c in aac aacd mp3h vorbis; do
  S=256; [ $c = vorbis ] && S=64
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_r05k_$c -o dec -- $REPO/symphonia_amd/build/decoders_bench --codec $c --streams $S --lookahead 256 --packets 512 --threads 16 > $OUT/r05k_dec_$c.log 2>&1
  echo "== decoders_bench --codec $c --streams $S --lookahead 256 --packets 512 --threads 16" >> $OUT/r05k_decoders_rocprofv3.txt
  tail -1 $OUT/r05k_dec_$c.log | cut -c1-400 >> $OUT/r05k_decoders_rocprofv3.txt
  python $REPO/tools/rocpd_summary.py $(find $OUT/prof_r05k_$c -name '*.db') 2>&1 | head -14 >> $OUT/r05k_decoders_rocprofv3.txt
  rm -rf $OUT/prof_r05k_$c
done
cd $REPO
timeout 600 python bench.py --selftest-multi 4 > $OUT/r05k_selftest_multi.json 2> $OUT/r05k_selftest_multi.err; echo "selftest rc=$?"
cut -c1-600 $OUT/r05k_selftest_multi.json
cut -c1-230 $OUT/r05k_decoders_rocprofv3.txt | head -60
