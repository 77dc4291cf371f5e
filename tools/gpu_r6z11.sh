#!/bin/bash
# Round 6: SQ / TA counters of the small TNS pass, lane by lane (LDSX 0) and through the LDS exchange (product)
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD
PASS_A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"
PASS_B="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"
PASS_C="TA_BUSY_avr TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
PASS_D="SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_CYCLES SQ_INSTS_VSKIPPED SQ_VALU_MFMA_BUSY_CYCLES"
cd /tmp
for v in product tns_ldsx0; do
  L=$REPO/build_ab/$v/libsymaccel.so; [ $v = product ] && L=$REPO/symphonia_amd/libsymaccel.so
  for p in A B C D; do
    eval "C=\$PASS_$p"
    SYMACCEL_LIB=$L timeout 200 rocprofv3 --kernel-trace --pmc $C -d $OUT/pmc_z11_${v}_$p -o x -- python $REPO/bench.py --workload aactns --steps 3 --warmup 1 --spinup-ms 40 --no-cpu-baseline --no-host-path --no-others --no-copy-ceiling > $OUT/pmc_z11_${v}_$p.log 2>&1
    echo "pmc $v pass $p rc=$?"
  done
  python $REPO/tools/rocpd_summary.py $(find $OUT/pmc_z11_${v}_A $OUT/pmc_z11_${v}_B $OUT/pmc_z11_${v}_C $OUT/pmc_z11_${v}_D -name '*.db') 2>&1 | grep -v 'at::native\|rocclr\|elementwise' > $OUT/r06z11_aactns_${v}_sq_counters.txt
  rm -rf $OUT/pmc_z11_${v}_?
  grep -A14 "tns_pair_kernel<12" $OUT/r06z11_aactns_${v}_sq_counters.txt | head -60
done
