cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/wino; mkdir -p $O; cd $R
TMF_STREAMS=1 TMF_ROOF_REPS=1 TMF_ROOF_SPIN_S=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/_p -o p -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also > $O/_p.log 2>&1; cp $O/_p/p_kernel_stats.csv $O/step_kernel_stats_1stream.csv; rm -rf $O/_p
head -40 $O/step_kernel_stats_1stream.csv | cut -c1-200
