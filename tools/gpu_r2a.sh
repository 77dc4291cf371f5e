#!/bin/bash
# Round-2 first GPU call: parity tests, bench lines of every workload, alternating A/B of prebuilt library variants.
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
rocminfo | grep -m3 -i "gfx\|Marketing" > $OUT/rocminfo_r2.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 2>&1 | tail -15 | tee $OUT/pytest_gpu_r2a.log
for w in aac mp3 vorbis alac; do
  timeout 300 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline > $OUT/r2a_bench_$w.json 2> $OUT/r2a_bench_$w.err
  echo "bench $w rc=$?"; tail -n 1 $OUT/r2a_bench_$w.json | cut -c1-260
done
timeout 600 python bench.py --workload flac --steps 10 --warmup 2 --no-cpu-baseline > $OUT/r2a_bench_flac.json 2> $OUT/r2a_bench_flac.err
echo "bench flac rc=$?"; tail -n 1 $OUT/r2a_bench_flac.json | cut -c1-260; tail -3 $OUT/r2a_bench_flac.err
ab() {  # workload, libs...
  W=$1; shift
  for rep in 1 2 3; do
    for lib in "$@"; do
      SYMACCEL_LIB=$lib timeout 120 python bench.py --workload $W --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W', '$(basename $lib)', 'ms_per_step', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],4))" | tee -a $OUT/r2a_ab.log
    done
  done
}
ab aac symphonia_amd/libsymaccel.so symphonia_amd/build/variants/lib_AAC_PREFETCH2.so symphonia_amd/build/variants/lib_AAC_NT1.so symphonia_amd/build/variants/lib_AAC_PREFETCH2_AAC_NT1.so
