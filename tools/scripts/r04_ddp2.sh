cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04k
rm -rf $O; mkdir -p $O
cd $R
timeout 600 python3 -m pytest tests/test_gpu_parallel.py -q -m gpu > $O/t_par.log 2>&1; echo "rc=$?" >> $O/t_par.log
tail -4 $O/t_par.log
for i in 1 2 3; do
python3 bench.py --no-also --no-cpu-baseline --steps 40 > $O/plain$i.json 2>> $O/err.log
TMF_DDP_FORCE=1 python3 bench.py --no-cpu-baseline --steps 40 > $O/ddp$i.json 2>> $O/err.log
done
python3 - <<P
import json
for n in ("plain1","ddp1","plain2","ddp2","plain3","ddp3"):
    try:
        d=json.loads(open("$O/"+n+".json").read().strip().splitlines()[-1]); print(n, d["value"], d["ms_per_step"], d.get("ms_per_step_min"), d.get("ms_per_step_median"), (d.get("per_rank") or {}).get("allreduce_exposed_ms_mean"), (d.get("per_rank") or {}).get("collective_kinds"))
    except Exception as e: print(n, "ERR", e)
P
TMF_DDP_FORCE=1 TMF_ROOF_REPS=1 TMF_ROOF_SPIN_S=0 TMF_BENCH_SETUP_STEPS=5 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/_p -o p -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/_p.log 2>&1
python3 tools/trace_gaps.py $O/_p/p_kernel_trace.csv --steps 8 --top 12 > $O/ddp_trace_gaps.txt 2>&1
head -16 $O/ddp_trace_gaps.txt | cut -c1-140
rm -rf $O/_p
