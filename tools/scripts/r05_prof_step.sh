cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; mkdir -p $O; cd $R
TMF_STREAMS=1 TMF_ROOF_REPS=1 TMF_ROOF_SPIN_S=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/_p -o p -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also > $O/_p.log 2>&1; cp $O/_p/p_kernel_stats.csv $O/r05_step_kernel_stats_1stream.csv; rm -rf $O/_p
TMF_ROOF_REPS=1 TMF_ROOF_SPIN_S=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/_p -o p -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also > $O/_p.log 2>&1; cp $O/_p/p_kernel_stats.csv $O/r05_step_kernel_stats.csv; python3 tools/trace_gaps.py $O/_p/p_kernel_trace.csv --steps 8 --top 16 > $O/r05_trace_gaps_fp32.txt 2>&1; rm -rf $O/_p
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/prof/r05_step_kernel_stats_1stream.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
steps = 43.0
print("sum kernel time per step (1 stream): %.3f ms" % (tot / steps / 1e6))
for r in rows[:34]:
    print("%-58s %5s  %8.1f us  %7.3f ms/step" % (r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:58], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / steps / 1e6))
PY
head -12 $O/r05_trace_gaps_fp32.txt
