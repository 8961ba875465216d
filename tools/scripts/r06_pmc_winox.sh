# counter passes over the split Winograd forward (conv2.0 at B = 8, 48^3), each its own process (--pmc only with --kernel-trace)
#   gpurun -- 'bash tools/scripts/r06_pmc_winox.sh'  -> gpurun_out/r06pmcx/summary.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06pmcx
rm -rf $O; mkdir -p $O
cd $R
run() { n=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $O/$n -o $n --output-format csv -- python3 tools/winox_one.py ${LAYER:-conv2.0} --reps 6 > $O/$n.log 2>&1
  python3 tools/pmc_summary.py $O/$n ${KERN:-winox_kernel} >> $O/summary.txt
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
run b SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run c SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT
cat $O/summary.txt
