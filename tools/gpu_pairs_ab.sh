#!/bin/bash
# Alternating A/B of prebuilt libraries on the Vorbis block-size pairs: bash tools/gpu_pairs_ab.sh <tag> "<pairs>" <reps> lib1.so lib2.so ...
# (PMC=1: one SQ counter pass per library and pair as well -- dynamic VALU / LDS / SALU instruction counts of the synthesis kernel)
TAG=$1; PAIRS=$2; REPS=$3; shift 3
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD
for rep in $(seq $REPS); do
  for lib in "$@"; do
    echo "## $(basename $lib)" | tee -a $OUT/${TAG}_pairs_ab.log
    SYMACCEL_LIB=$REPO/$lib timeout 300 python tools/vorbis_pairs_probe.py $PAIRS 2>&1 | grep -v "^ROCm\|^Hostname\|^Librccl\|^RCCL\|^HIP\|amdgpu.ids" | tee -a $OUT/${TAG}_pairs_ab.log
  done
done
if [ -n "$PMC" ]; then
  cd /tmp
  for lib in "$@"; do
    for pr in $PAIRS; do
      n=$(basename $lib .so)_${pr/,/_}
      SYMACCEL_LIB=$REPO/$lib timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES -d $OUT/pmcab_$n -o x -- python $REPO/tools/vorbis_pairs_probe.py $pr > /dev/null 2>&1
      python $REPO/tools/rocpd_summary.py $(find $OUT/pmcab_$n -name '*.db') 2>&1 | grep "vorbis_synth" | awk -v n=$n '{ if ($0 ~ /SQ_/) print n, $(NF-4), $(NF-2); else print n, "avg_us", $(NF-10) }' | tee -a $OUT/${TAG}_pairs_pmc.log
      rm -rf $OUT/pmcab_$n
    done
  done
fi
