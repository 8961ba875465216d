# counter passes over the first block's forward / one-pass backward (B = 8, 96^3), each its own process (--pmc only with --kernel-trace)
#   gpurun -- 'bash tools/scripts/r06_pmc_c1.sh'  -> gpurun_out/r06pmcc1/summary.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06pmcc1
rm -rf $O; mkdir -p $O
cd $R
run() { n=$1; w=$2; shift 2
  timeout 200 rocprofv3 --kernel-trace --pmc "$@" -d $O/$n -o $n --output-format csv -- python3 tools/c1_one.py $w --reps 6 > $O/$n.log 2>&1
  echo "## $w: $@" >> $O/summary.txt
  python3 tools/pmc_summary.py $O/$n conv1_fused_kernel >> $O/summary.txt
  rm -rf $O/$n
}
for w in fwd bwd; do
run a_$w $w SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
run b_$w $w SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run c_$w $w SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVES
done
cat $O/summary.txt
