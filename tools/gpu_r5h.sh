#!/bin/bash
# Round 5: the T1M exchange in the short-window transform (Vorbis 256 / 2048 short runs, AAC EIGHT_SHORT) + the AAC_DECODE batch kind on the GPU
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD
timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 2>&1 | grep -v -E "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -n 8 > $OUT/r05h_gputest.log
cat $OUT/r05h_gputest.log
for rep in 1 2; do
for w in "vorbis" "aac --aac-mix 0.25" "aac"; do
  timeout 300 python bench.py --workload $w --no-others --no-cpu-baseline --no-copy-ceiling --no-host-path --repeats 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', round(d['ms_per_step'],4), round(d['roofline']['frac'],4), d['repeats']['ms_per_step_median'])"
done
done | tee $OUT/r05h_t1m.txt
bash tools/gpu_pmc.sh r05h vorbis
grep "LDS_BANK\|LDS_IDX\|avg_us\|vorbis_synth_wave_kernel<0>(symacce.*calls" $OUT/r05h_vorbis_sq_counters.txt | cut -c1-40,88-190 | head
grep "wave_kernel" $OUT/r05h_vorbis_sq_counters.txt | cut -c1-40,88-190 | head -20
