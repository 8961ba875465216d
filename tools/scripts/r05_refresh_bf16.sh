# re-run of the bf16 lines of refresh_profiles.sh after the first-block pair sums were restricted to the fp32 entry
TAG=r05
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/refresh2
rm -rf $O; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -4 > $O/gputest.txt
timeout 900 python3 bench.py > $O/b.log 2>&1; grep "^{" $O/b.log | tail -1 > $O/${TAG}_bench_n1.json
timeout 600 python3 bench.py --precision bf16 --storage bf16 --size 128 > $O/b.log 2>&1; grep "^{" $O/b.log | tail -1 > $O/${TAG}_bench_128_bf16_storage.json
timeout 300 python3 bench.py --precision bf16 --size 128 --no-cpu-baseline > $O/b.log 2>&1; grep "^{" $O/b.log | tail -1 > $O/${TAG}_bench_128_bf16.json
timeout 300 python3 bench.py --precision bf16 --storage bf16 --no-cpu-baseline > $O/b.log 2>&1; grep "^{" $O/b.log | tail -1 > $O/${TAG}_bench_96_bf16_storage.json
timeout 300 python3 bench.py --eval --precision bf16 --storage bf16 --size 128 --no-cpu-baseline > $O/b.log 2>&1; grep "^{" $O/b.log | tail -1 > $O/${TAG}_bench_eval_128_bf16_storage.json
prof() { n=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/_p -o p -- python3 bench.py "$@" > $O/_p.log 2>&1
  cp $O/_p/p_kernel_stats.csv $O/${TAG}_${n}_kernel_stats.csv; grep "^{" $O/_p.log | tail -1 > $O/${TAG}_${n}.json; rm -rf $O/_p
}
prof roofline_only_128_bf16_storage --roofline-only --no-cpu-baseline --precision bf16 --storage bf16 --size 128
TMF_STREAMS=1 TMF_ROOF_REPS=1 TMF_ROOF_SPIN_S=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/_p -o p -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --precision bf16 --storage bf16 --size 128 > $O/_p.log 2>&1; cp $O/_p/p_kernel_stats.csv $O/${TAG}_bench_128_bf16_storage_kernel_stats_1stream.csv; rm -rf $O/_p
TMF_ROOF_REPS=1 TMF_ROOF_SPIN_S=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/_p -o p -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --precision bf16 --storage bf16 --size 128 > $O/_p.log 2>&1; python3 tools/trace_gaps.py $O/_p/p_kernel_trace.csv --steps 8 --top 30 > $O/${TAG}_trace_gaps_128_bf16_storage.txt 2>&1; rm -rf $O/_p
rm -f $O/b.log $O/_p.log
cat $O/gputest.txt; python3 - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], d.get("value"), d.get("ms_per_step"))
    except Exception as e: print(f, e)
PY
