# PMC passes of the 8x8x8-brick bf16 forward kernel (bf16 tensors, LDS-DMA form) on conv2.0 (NT = 1) and conv2.3 (NT = 2)
# at 64^3 (configs[2] shapes): where do the launch's cycles go?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc16r3
rm -rf $O; mkdir -p $O
run() { # name layer pmc...
  n=$1; l=$2; shift 2
  timeout 120 rocprofv3 --kernel-trace --pmc "$@" -d $O/$n -o $n --output-format csv -- python3 $R/tools/kone.py bf16s $l --S 128 --reps 6 --v2 1 > $O/$n.log 2>&1
}
for l in conv2.0 conv2.3; do
  run ${l}_p1 $l SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_WAVES
  run ${l}_p2 $l SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
  run ${l}_p3 $l SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA
  run ${l}_p4 $l SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL
  run ${l}_p5 $l TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum
  run ${l}_p6 $l TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum TCC_BUSY_avr
  run ${l}_p7 $l SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_COEXEC_CYCLES
done
cd $R
for l in conv2.0 conv2.3; do
  for p in p1 p2 p3 p4 p5 p6 p7; do echo "== $l $p"; python3 tools/pmc_summary.py $O/${l}_$p conv3d_fwd_bf16; tail -2 $O/${l}_$p.log | grep -i "error\|fail"; done
done > gpurun_out/pmc16r3_summary.txt 2>&1
cat gpurun_out/pmc16r3_summary.txt
