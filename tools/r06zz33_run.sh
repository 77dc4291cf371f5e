# FLAC config 5 on padded rows: the pitch at order 32 (bench data), then build variants at the product's pitch
OUT=gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do
for pitch in 4096 4352 4608 4736 4864 5120 5632 6144; do
  SYM_BENCH_PITCH=$pitch timeout 120 python bench.py --workload flacp --steps 20 --warmup 3 --no-cpu-baseline --no-others --no-host-path --no-copy-ceiling 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('flacp pitch $pitch', 'ms_per_step', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],4), 'mismatches', d['verified']['mismatches'])" | tee -a $OUT/r06zz33_flac_pitch.txt
done; done
STEPS=20 WARMUP=3 bash tools/gpu_ab_libs.sh r06zz33 flacp 2 symphonia_amd/libsymaccel.so build_ab/flac_w3.so build_ab/flac_sw.so build_ab/flac_g2.so
for pitch in 4352 4608 4864 5120; do
  SYM_BENCH_PITCH=$pitch timeout 120 python bench.py --workload alacp --steps 20 --warmup 3 --no-cpu-baseline --no-others --no-host-path --no-copy-ceiling 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('alacp pitch $pitch', 'ms_per_step', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],4), 'mismatches', d['verified']['mismatches'])" | tee -a $OUT/r06zz33_flac_pitch.txt
done
