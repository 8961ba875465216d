cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04j
rm -rf $O; mkdir -p $O
cd $R
TMF_DDP_FORCE=1 TMF_ROOF_REPS=1 TMF_ROOF_SPIN_S=0 TMF_BENCH_SETUP_STEPS=5 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/_p -o p -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/_p.log 2>&1
cp $O/_p/p_kernel_stats.csv $O/ddp_kernel_stats.csv
python3 tools/trace_gaps.py $O/_p/p_kernel_trace.csv --steps 8 --top 30 > $O/ddp_trace_gaps.txt 2>&1
grep -i "nccl\|rccl\|AllReduce\|div\|Functor" $O/ddp_kernel_stats.csv | head -20
head -30 $O/ddp_trace_gaps.txt
rm -rf $O/_p
