#!/bin/bash
# Round 5, second GPU call: the cross-stream batcher on the GPU (tests), the trait-level decoders line, rocprofv3 kernel stats of it.
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD
timeout 600 python -m pytest tests/test_batcher.py tests/test_lookahead.py tests/test_abi.py -m gpu -q -x --timeout 300 2>&1 | grep -v -E "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -n 12 > $OUT/r05b_gputest.log
cat $OUT/r05b_gputest.log
timeout 900 python bench.py --workload decoders > $OUT/r05b_decoders.json 2> $OUT/r05b_decoders.err
echo "decoders rc=$?"; cut -c1-1500 $OUT/r05b_decoders.json; tail -5 $OUT/r05b_decoders.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_r05b_dec -o dec -- $REPO/symphonia_amd/build/decoders_bench --codec aac --streams 256 --lookahead 64 --packets 256 --threads 16 > $OUT/r05b_dec_prof.log 2>&1
python $REPO/tools/rocpd_summary.py $(find $OUT/prof_r05b_dec -name '*.db') > $OUT/r05b_decoders_rocprofv3.txt 2>&1
rm -rf $OUT/prof_r05b_dec
head -20 $OUT/r05b_decoders_rocprofv3.txt
for T in 1 4 16 32; do $REPO/symphonia_amd/build/decoders_bench --codec aac --streams 256 --lookahead 64 --packets 256 --threads $T; done > $OUT/r05b_threads_sweep.txt 2>&1
for L in 8 16 32 128 256; do $REPO/symphonia_amd/build/decoders_bench --codec aac --streams 256 --lookahead $L --packets 512 --threads 16; done >> $OUT/r05b_threads_sweep.txt 2>&1
cut -c1-330 $OUT/r05b_threads_sweep.txt
nproc; lscpu | grep "Model name"
