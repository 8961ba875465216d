cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmctcc
rm -rf $O; mkdir -p $O
run() { n=$1; l=$2; shift 2
  timeout 120 rocprofv3 --kernel-trace --pmc "$@" -d $O/$n -o $n --output-format csv -- python3 $R/tools/kone.py bf16s $l --S 128 --reps 6 --v2 1 > $O/$n.log 2>&1
}
for l in conv2.0 conv2.3; do
  run ${l}_a $l TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum
  run ${l}_b $l TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
  run ${l}_c $l FETCH_SIZE WRITE_SIZE
done
cd $R
for l in conv2.0 conv2.3; do for p in a b c; do echo "== $l $p"; python3 tools/pmc_summary.py $O/${l}_$p conv3d_fwd_bf16; done; done
