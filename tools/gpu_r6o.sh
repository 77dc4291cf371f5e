#!/bin/bash
# Round 6: the floor-1 kernel's 174 us (vorbisf: long + short launch) attributed to its phases: ablation builds (SYMACCEL_TUNE_F1_ABLATE,
# results wrong on purpose) and workgroup shapes (SYMACCEL_TUNE_F1_WAVES) under rocprofv3 --kernel-trace
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD
cd /tmp
: > $OUT/r06o_floor1_phases.txt
for lib in symphonia_amd/libsymaccel.so $(ls $REPO/build_ab/f1*.so | sed "s#$REPO/##"); do
    n=$(basename $lib .so)
    SYMACCEL_LIB=$REPO/$lib timeout 200 rocprofv3 --kernel-trace -d $OUT/kt_r06o_$n -o p -- python $REPO/bench.py --workload vorbisf --steps 10 --warmup 2 --no-spinup --repeats 0 --no-verify --no-cpu-baseline --no-host-path --no-others --no-copy-ceiling > $OUT/kt_r06o_$n.log 2>&1
    echo "== $n rc=$?" | tee -a $OUT/r06o_floor1_phases.txt
    python $REPO/tools/rocpd_summary.py $(find $OUT/kt_r06o_$n -name '*.db') 2>&1 | grep -E "floor1|synth_wave" | head -6 >> $OUT/r06o_floor1_phases.txt
    rm -rf $OUT/kt_r06o_$n
done
cd $REPO
for lib in symphonia_amd/libsymaccel.so build_ab/f1w2.so build_ab/f1w8.so; do
    SYMACCEL_LIB=$PWD/$lib timeout 120 python bench.py --workload vorbisf --steps 100 --warmup 10 --repeats 0 --no-cpu-baseline --no-host-path --no-others --no-copy-ceiling 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('vorbisf', '$(basename $lib)', 'ms_per_step', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],4), 'verify', d.get('verify'))" | tee -a $OUT/r06o_floor1_phases.txt
done
