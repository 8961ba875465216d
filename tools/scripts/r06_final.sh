# the contract line and its roofline evidence once more with the final bench.py (first-block rows priced on the bf16 pipe) + the counter passes
#   gpurun --timeout 2400 -- 'bash tools/scripts/r06_final.sh'  -> gpurun_out/final/*, gpurun_out/r06pmc/*
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
cd $R
TAG=r06
timeout 600 python3 bench.py > $O/b.log 2>&1; grep "^{" $O/b.log | tail -1 > $O/${TAG}_bench_n1.json
timeout 300 python3 bench.py --no-also --no-cpu-baseline --steps 40 > $O/b.log 2>&1; grep "^{" $O/b.log | tail -1 > $O/${TAG}_bench_n1_plain40.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/_p -o p -- python3 bench.py --roofline-only --no-cpu-baseline > $O/_p.log 2>&1
cp $O/_p/p_kernel_stats.csv $O/${TAG}_roofline_only_kernel_stats.csv; grep "^{" $O/_p.log | tail -1 > $O/${TAG}_roofline_only.json; rm -rf $O/_p
rm -f $O/b.log $O/_p.log
bash tools/scripts/r06_pmc.sh > $O/pmc.log 2>&1
tail -30 $O/pmc.log
ls -la $O $R/gpurun_out/r06pmc
