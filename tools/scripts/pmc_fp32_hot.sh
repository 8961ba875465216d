cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc32
mkdir -p $O
run() { n=$1; what=$2; shift 2
  timeout 200 rocprofv3 --kernel-trace --pmc "$@" -d $O/$n -o $n --output-format csv -- python3 $R/tools/kone.py $what conv2.3 --S 96 --reps 12 --spin 700 > $O/$n.log 2>&1
}
for what in fwd wgrad; do
  run ${what}_p1 $what SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_WAVES
  run ${what}_p2 $what SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_LDS
done
run fwd_p3 fwd FETCH_SIZE
run fwd_p4 fwd WRITE_SIZE
run fwd_p5 fwd TCC_HIT_sum TCC_MISS_sum
cd $R
for n in fwd_p1 fwd_p2 fwd_p3 fwd_p4 fwd_p5; do echo "== $n"; python3 tools/pmc_summary.py $O/$n "FwdCfg<3, 16, 1, 2" --skip 4; done > gpurun_out/pmc32_summary.txt 2>&1
for n in wgrad_p1 wgrad_p2; do echo "== $n"; python3 tools/pmc_summary.py $O/$n conv3d_wgrad_kernel --skip 4; done >> gpurun_out/pmc32_summary.txt 2>&1
