#!/bin/bash
# Round 6: both floor-1 classes of the vorbisf workload in one grid (vorbis_floor1_pair_kernel): parity on the GPU, the bench line, the kernel trace
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD
timeout 900 python -m pytest tests/test_vorbis_floor_y.py tests/test_vorbis_decode.py tests/test_batcher_kinds.py tests/test_lookahead.py -x -q -m gpu 2>&1 | tail -5 | tee $OUT/r06p_gputest.txt
for i in 1 2; do
timeout 200 python bench.py --workload vorbisf --steps 200 --warmup 20 --no-cpu-baseline --no-host-path --no-others --no-copy-ceiling 2>/dev/null > $OUT/r06p_bench_vorbisf_$i.json
python -c "import json; d=json.load(open('$OUT/r06p_bench_vorbisf_$i.json')); print('vorbisf ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'verified', d.get('verified'))" | tee -a $OUT/r06p_gputest.txt
done
cd /tmp
timeout 200 rocprofv3 --kernel-trace -d $OUT/kt_r06p -o p -- python $REPO/bench.py --workload vorbisf --steps 20 --warmup 2 --no-spinup --repeats 0 --no-cpu-baseline --no-host-path --no-others --no-copy-ceiling > $OUT/kt_r06p.log 2>&1
python $REPO/tools/rocpd_summary.py $(find $OUT/kt_r06p -name '*.db') 2>&1 | head -12 | tee $OUT/r06p_vorbisf_rocprofv3.txt
rm -rf $OUT/kt_r06p
