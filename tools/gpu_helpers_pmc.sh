#!/bin/bash
# (every pass under its own `timeout`: rocprofv3 has died at start-up on a fresh box twice and then sat until the gpurun limit)
# SQ / LDS / L1-L2 counter passes over tools/bench_helpers.py, filtered to the kernels matching <pattern>:
#   bash tools/gpu_helpers_pmc.sh <tag> <pattern>      -> gpurun_out/<tag>_helper_counters.txt
TAG=$1; PAT=$2
export TMPDIR=/tmp; OUT=$PWD/gpurun_out; REPO=$PWD; mkdir -p $OUT; cd /tmp
timeout 150 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU -d $OUT/pmc_${TAG}_A -o h -- python $REPO/tools/bench_helpers.py > $OUT/pmc_${TAG}_A.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES -d $OUT/pmc_${TAG}_B -o h -- python $REPO/tools/bench_helpers.py > $OUT/pmc_${TAG}_B.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum -d $OUT/pmc_${TAG}_C -o h -- python $REPO/tools/bench_helpers.py > $OUT/pmc_${TAG}_C.log 2>&1
cd $REPO
python tools/rocpd_summary.py gpurun_out/pmc_${TAG}_A/h_results.db gpurun_out/pmc_${TAG}_B/h_results.db gpurun_out/pmc_${TAG}_C/h_results.db | grep -i "$PAT" > gpurun_out/${TAG}_helper_counters.txt
rm -rf gpurun_out/pmc_${TAG}_A gpurun_out/pmc_${TAG}_B gpurun_out/pmc_${TAG}_C
