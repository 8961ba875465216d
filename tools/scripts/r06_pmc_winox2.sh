# second counter set over the split Winograd forward (queues, instruction fetch, co-execution, texture / L1 side)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06pmcx2
rm -rf $O; mkdir -p $O
cd $R
run() { n=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $O/$n -o $n --output-format csv -- python3 tools/winox_one.py ${LAYER:-conv2.0} --reps 6 > $O/$n.log 2>&1
  python3 tools/pmc_summary.py $O/$n ${KERN:-winox_kernel} >> $O/summary.txt
}
run a SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_IFETCH_LEVEL
run b SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU2 SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_CYCLES
run d TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
cat $O/summary.txt
