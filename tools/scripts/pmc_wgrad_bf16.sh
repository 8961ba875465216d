cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmcwg
rm -rf $O; mkdir -p $O
run() { n=$1; what=$2; shift 2
  timeout 120 rocprofv3 --kernel-trace --pmc "$@" -d $O/$n -o $n --output-format csv -- python3 $R/tools/kone.py $what conv2.3 --S 128 --reps 6 > $O/$n.log 2>&1
}
run p1 wgrad16s SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_WAVES
run p2 wgrad16s SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_LDS
run p3 wgrad16s SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run p4 wgrad16s TCC_HIT_sum TCC_MISS_sum
run p5 wgrad16s FETCH_SIZE
run p6 wgrad16s WRITE_SIZE
cd $R
for p in p1 p2 p3 p4 p5 p6; do python3 tools/pmc_summary.py $O/$p conv3d_wgrad_bf16; done
