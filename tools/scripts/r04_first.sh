# round 4, first pass: the GPU suite, the default bench line (headline + `also`), the 1-rank DDP overhead
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04a
rm -rf $O; mkdir -p $O
cd $R
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python3 - <<P
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"], d.get("ms_per_step_min"), d.get("ms_per_step_median"), "cpu", d["cpu_baseline"]["value"])
for a in d.get("also", []):
    print(" also:", a.get("config_name"), a.get("value"), a.get("ms_per_step"), a.get("error"), (a.get("cpu_baseline") or {}).get("value"))
P
python3 bench.py --no-also --no-cpu-baseline --steps 40 > $O/bench_plain.json 2>> $O/bench_default.err
TMF_DDP_FORCE=1 python3 bench.py --no-also --no-cpu-baseline --steps 40 > $O/bench_ddp1.json 2> $O/bench_ddp1.err
python3 bench.py --no-also --no-cpu-baseline --steps 40 > $O/bench_plain2.json 2>> $O/bench_default.err
TMF_DDP_FORCE=1 python3 bench.py --no-also --no-cpu-baseline --steps 40 > $O/bench_ddp1b.json 2>> $O/bench_ddp1.err
python3 - <<P
import json
for n in ("bench_plain","bench_ddp1","bench_plain2","bench_ddp1b"):
    try:
        d=json.loads(open("$O/"+n+".json").read().strip().splitlines()[-1]); print(n, d["value"], d["ms_per_step"], d.get("ms_per_step_median"), (d.get("per_rank") or {}).get("allreduce_exposed_ms_mean"), (d.get("per_rank") or {}).get("collective_kinds"))
    except Exception as e: print(n, "ERR", e)
P
timeout 3000 python3 -m pytest tests/ -q -m gpu -x > $O/t_all.log 2>&1; echo "rc=$?" >> $O/t_all.log
tail -15 $O/t_all.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
