# PMC passes of the bf16 forward kernel on conv2.3 at 64^3 (configs[2] shape), bf16 tensors, both kernel variants.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc16v2
mkdir -p $O
run() { # name v2mode pmc...
  n=$1; v=$2; shift 2
  timeout 120 rocprofv3 --kernel-trace --pmc "$@" -d $O/$n -o $n --output-format csv -- python3 $R/tools/kone.py bf16s conv2.3 --S 128 --reps 6 --v2 $v > $O/$n.log 2>&1
}
for v in 0 2; do
  run v${v}_p1 $v SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_WAVES
  run v${v}_p2 $v SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_LDS
  run v${v}_p3 $v SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
  run v${v}_p4 $v FETCH_SIZE
  run v${v}_p5 $v WRITE_SIZE
done
cd $R
for v in 0 2; do
  for p in p1 p2 p3 p4 p5; do echo "== v$v $p"; python3 tools/pmc_summary.py $O/v${v}_$p conv3d_fwd_bf16; done
done > gpurun_out/pmc16v2_summary.txt 2>&1
cat gpurun_out/pmc16v2_summary.txt
