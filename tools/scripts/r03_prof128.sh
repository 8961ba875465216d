cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/p128
rm -rf $O; mkdir -p $O
cd $R
TMF_STREAMS=1 TMF_ROOF_REPS=1 TMF_ROOF_SPIN_S=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/_p -o p -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --precision bf16 --storage bf16 --size 128 > $O/_p.log 2>&1; cp $O/_p/p_kernel_stats.csv $O/stats_1stream.csv; rm -rf $O/_p
TMF_ROOF_REPS=1 TMF_ROOF_SPIN_S=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/_p -o p -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --precision bf16 --storage bf16 --size 128 > $O/_p.log 2>&1; python3 tools/trace_gaps.py $O/_p/p_kernel_trace.csv --steps 8 --top 30 > $O/gaps.txt 2>&1; rm -rf $O/_p
