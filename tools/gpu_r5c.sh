#!/bin/bash
# Round 5, third GPU call: the full GPU suite on the batcher rewrite (slab slots + gather / scatter kernels) and the 4096 / 8192
# Vorbis pair on the workgroup kernel; the decoders line; the block-size pairs.
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD
timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 2>&1 | grep -v -E "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -n 12 > $OUT/r05c_gputest.log
cat $OUT/r05c_gputest.log
for T in 1 4 16 32; do $REPO/symphonia_amd/build/decoders_bench --codec aac --streams 256 --lookahead 64 --packets 256 --threads $T; done > $OUT/r05c_threads_sweep.txt 2>&1
for L in 16 256; do $REPO/symphonia_amd/build/decoders_bench --codec aac --streams 256 --lookahead $L --packets 512 --threads 16; done >> $OUT/r05c_threads_sweep.txt 2>&1
$REPO/symphonia_amd/build/decoders_bench --codec aac --streams 1024 --lookahead 64 --packets 128 --threads 32 >> $OUT/r05c_threads_sweep.txt 2>&1
cut -c1-420 $OUT/r05c_threads_sweep.txt
timeout 900 python bench.py --workload decoders > $OUT/r05c_decoders.json 2> $OUT/r05c_decoders.err
echo "decoders rc=$?"; tail -3 $OUT/r05c_decoders.err
timeout 300 python tools/vorbis_pairs_probe.py 12,13 10,13 9,12 8,11 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP\|amdgpu.ids" | tee $OUT/r05c_vorbis_pairs.txt
