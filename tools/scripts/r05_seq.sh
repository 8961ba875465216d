cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; mkdir -p $O; cd $R
TMF_ROOF_REPS=1 TMF_ROOF_SPIN_S=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/_p -o p -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also > $O/_p.log 2>&1
python3 tools/trace_seq.py $O/_p/p_kernel_trace.csv --from-ms 2.3 --to-ms 4.6 > $O/r05_trace_seq_mid.txt 2>&1
python3 tools/trace_seq.py $O/_p/p_kernel_trace.csv > $O/r05_trace_seq_all.txt 2>&1
rm -rf $O/_p
cat $O/r05_trace_seq_mid.txt
