#!/bin/bash
# SQ counter passes of the Vorbis kernels on single block-size pairs: bash tools/gpu_pairs_pmc.sh <tag> 7,10 9,12 ...
TAG=$1; shift
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD
cd /tmp
PASS_A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"
PASS_B="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES"
for pr in "$@"; do
  n=${pr/,/_}
  for p in A B; do
    eval "C=\$PASS_$p"
    timeout 200 rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_${TAG}_${n}_$p -o p$n -- python $REPO/tools/vorbis_pairs_probe.py $pr > $OUT/pmc_${TAG}_${n}_$p.log 2>&1
    echo "pmc $pr pass $p rc=$?"
  done
  python $REPO/tools/rocpd_summary.py $(find $OUT/pmc_${TAG}_${n}_A $OUT/pmc_${TAG}_${n}_B -name '*.db') 2>&1 | grep -v 'at::native\|rocclr\|state_copy\|offsets_kernel' > $OUT/${TAG}_pairs_${n}_sq.txt
  rm -rf $OUT/pmc_${TAG}_${n}_A $OUT/pmc_${TAG}_${n}_B
done
