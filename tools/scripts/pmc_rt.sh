# PMC passes of the register-tiled fp32 forward kernel (conv3.0 at 24^3, B=8) and of the ring kernel on the same layer
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmcrt
rm -rf $O; mkdir -p $O
run() { n=$1; rt=$2; shift 2
  TMF_CONV_RT=$rt timeout 120 rocprofv3 --kernel-trace --pmc "$@" -d $O/$n -o $n --output-format csv -- python3 $R/tools/kone.py fwd conv3.0 --S 96 --reps 8 --spin 20 > $O/$n.log 2>&1
}
for rt in 1 0; do
  run rt${rt}_p1 $rt SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_WAVES
  run rt${rt}_p2 $rt SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_INSTS_MFMA
  run rt${rt}_p3 $rt SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
done
cd $R
for rt in 1 0; do for p in p1 p2 p3; do echo "== rt=$rt $p"; python3 tools/pmc_summary.py $O/rt${rt}_$p "conv3d_fwd" --skip 21; done; done
