cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc16
mkdir -p $O
run() { # name what layer S pmc...
  n=$1; what=$2; S=$3; shift 3
  timeout 120 rocprofv3 --kernel-trace --pmc "$@" -d $O/$n -o $n --output-format csv -- python3 $R/tools/kone.py $what conv2.3 --S $S --reps 6 > $O/$n.log 2>&1
}
for what in bf16 wgrad16 bf16s wgrad16s; do
  run ${what}_p1 $what 128 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_WAVES
  run ${what}_p2 $what 128 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_LDS
  run ${what}_p3 $what 128 FETCH_SIZE
  run ${what}_p4 $what 128 WRITE_SIZE
done
cd $R
for what in bf16 wgrad16 bf16s wgrad16s; do
  k=conv3d_fwd_bf16_kernel; [ $what = wgrad16 ] && k=conv3d_wgrad_bf16_kernel; [ $what = wgrad16s ] && k=conv3d_wgrad_bf16_kernel
  for p in p1 p2 p3 p4; do echo "== $what $p"; python3 tools/pmc_summary.py $O/${what}_$p $k; done
done > gpurun_out/pmc16_summary.txt 2>&1
