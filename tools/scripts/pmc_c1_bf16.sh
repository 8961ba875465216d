# PMC passes over the four bf16 passes of the fused first block at 128^3 (tools/kc1.py --S 128 --bf16).
#   gpurun -- 'bash tools/scripts/pmc_c1_bf16.sh'   -> gpurun_out/pmcc1/*.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmcc1
rm -rf $O; mkdir -p $O
run() { n=$1; shift
  timeout 120 rocprofv3 --kernel-trace --pmc "$@" -d $O/$n -o $n --output-format csv -- python3 $R/tools/kc1.py --S 128 --bf16 > $O/$n.log 2>&1
}
run p1 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_WAVES
run p2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_LDS
run p3 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run p4 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU
cd $R
for m in 0 1 2 3; do for p in p1 p2 p3 p4; do echo "== pass $m $p"; python3 tools/pmc_summary.py $O/$p "conv1_fused_kernel<$m"; done; done > $O/summary.txt 2>&1
cat $O/summary.txt
