#!/bin/bash
# SQ counter passes for the synthesis kernels.  bash tools/gpu_pmc.sh <tag> <workload> [more workloads]
TAG=$1; shift
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD
cd /tmp
[ -f $OUT/counters.txt ] || rocprofv3 -L > $OUT/counters.txt 2>&1
PASS_A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"
PASS_B="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES"
for w in "$@"; do
  for p in A B; do
    eval "C=\$PASS_$p"
    timeout 200 rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_${TAG}_${w}_$p -o $w -- python $REPO/bench.py --workload $w --steps 3 --warmup 1 --spinup-ms 40 --no-cpu-baseline --no-host-path --no-others --no-copy-ceiling > $OUT/pmc_${TAG}_${w}_$p.log 2>&1
    echo "pmc $w pass $p rc=$?"
  done
done
# text summaries (kernels of the workload only) next to the databases, then drop the databases (too large to copy back)
for w in "$@"; do
  python $REPO/tools/rocpd_summary.py $(find $OUT/pmc_${TAG}_${w}_A $OUT/pmc_${TAG}_${w}_B -name '*.db') 2>&1 | grep -v 'at::native\|rocclr\|elementwise' > $OUT/${TAG}_${w}_sq_counters.txt
  rm -rf $OUT/pmc_${TAG}_${w}_A $OUT/pmc_${TAG}_${w}_B
done
