#!/bin/bash
# Round 5, first GPU call: the GPU suite on the new entry points, the default bench line (repeats / median), SQ-counter passes of the
# three kernels that had none (mp3_synth_kernel<4,true>, alac, flac).   bash tools/gpu_r5a.sh
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD
timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 2>&1 | grep -v -E "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -n 8 > $OUT/r05a_gputest.log
cat $OUT/r05a_gputest.log
timeout 600 python bench.py > $OUT/r05a_default_bench.json 2> $OUT/r05a_default_bench.err
echo "default bench rc=$?"; cut -c1-400 $OUT/r05a_default_bench.json
bash tools/gpu_pmc.sh r05a mp3q alac flac
ls $OUT | head -50
