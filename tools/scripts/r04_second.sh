cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04b
rm -rf $O; mkdir -p $O
cd $R
timeout 3000 python3 -m pytest tests/ -q -m gpu > $O/t_all.log 2>&1; echo "rc=$?" >> $O/t_all.log
tail -12 $O/t_all.log
for i in 1 2; do
python3 bench.py --no-also --no-cpu-baseline --steps 40 > $O/bench_plain$i.json 2>> $O/bench.err
TMF_DDP_FORCE=1 python3 bench.py --no-also --no-cpu-baseline --steps 40 > $O/bench_ddp$i.json 2>> $O/bench_ddp.err
done
TMF_DDP_FORCE=1 TMF_DDP_INPLACE=0 python3 bench.py --no-also --no-cpu-baseline --steps 40 > $O/bench_ddp_buckets.json 2>> $O/bench_ddp.err
python3 - <<P
import json
for n in ("bench_plain1","bench_ddp1","bench_plain2","bench_ddp2","bench_ddp_buckets"):
    try:
        d=json.loads(open("$O/"+n+".json").read().strip().splitlines()[-1]); print(n, d["value"], d["ms_per_step"], d.get("ms_per_step_median"), (d.get("per_rank") or {}).get("allreduce_exposed_ms_mean"), (d.get("per_rank") or {}).get("collective_kinds"))
    except Exception as e: print(n, "ERR", e)
P
tail -5 $O/bench_ddp.err
