#!/bin/bash
# Round 6: SQ_LDS_BANK_CONFLICT of the Vorbis 256 / 2048 walk attributed to LDS phases: ablation builds (SYMACCEL_TUNE_LDS_ABLATE, results wrong
# on purpose) under rocprofv3 --pmc; workloads vorbis (f32 spectra, config-4 shard) and vorbisf (residue + byte plane)
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD
cd /tmp
C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES"
: > $OUT/r06n_lds_phases.txt
for lib in symphonia_amd/libsymaccel.so $(ls $REPO/build_ab/abl*.so | sed "s#$REPO/##"); do
  for w in vorbis vorbisf; do
    n=$(basename $lib .so)_$w
    SYMACCEL_LIB=$REPO/$lib timeout 200 rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_r06n_$n -o p -- python $REPO/bench.py --workload $w --steps 5 --warmup 1 --no-spinup --repeats 0 --no-verify --no-cpu-baseline --no-host-path --no-others --no-copy-ceiling > $OUT/pmc_r06n_$n.log 2>&1
    echo "== $n rc=$?" | tee -a $OUT/r06n_lds_phases.txt
    python $REPO/tools/rocpd_summary.py $(find $OUT/pmc_r06n_$n -name '*.db') 2>&1 | grep -A12 "vorbis_synth_wave_kernel" | head -16 >> $OUT/r06n_lds_phases.txt
    rm -rf $OUT/pmc_r06n_$n
  done
done
# and the timings (no counters)
cd $REPO
for lib in symphonia_amd/libsymaccel.so $(ls build_ab/abl*.so); do
  for w in vorbis vorbisf; do
    SYMACCEL_LIB=$PWD/$lib timeout 120 python bench.py --workload $w --steps 100 --warmup 10 --repeats 0 --no-verify --no-cpu-baseline --no-host-path --no-others --no-copy-ceiling 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', '$(basename $lib)', 'ms_per_step', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],4))" | tee -a $OUT/r06n_lds_phases.txt
  done
done
