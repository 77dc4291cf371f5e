export TMPDIR=/tmp; OUT=$PWD/gpurun_out; REPO=$PWD; cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $OUT/pmc_tns_A -o h -- python $REPO/tools/bench_helpers.py > $OUT/pmc_tns_A.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TA_BUSY_avr TA_TA_BUSY_sum -d $OUT/pmc_tns_B -o h -- python $REPO/tools/bench_helpers.py > $OUT/pmc_tns_B.log 2>&1
cd $REPO
python tools/rocpd_summary.py gpurun_out/pmc_tns_A/h_results.db gpurun_out/pmc_tns_B/h_results.db | grep -i "tns\|floor1" > gpurun_out/r2zc_tns_counters.txt
rm -rf gpurun_out/pmc_tns_A gpurun_out/pmc_tns_B
