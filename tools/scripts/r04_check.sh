cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04z
rm -rf $O; mkdir -p $O
cd $R
timeout 3000 python3 -m pytest tests/ -q -m gpu > $O/t_all.log 2>&1; echo "rc=$?" >> $O/t_all.log
tail -6 $O/t_all.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python3 - <<P
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"], d.get("ms_per_step_min"), d.get("ms_per_step_median"), "cpu", d["cpu_baseline"]["value"], d["roofline"]["frac"])
for a in d.get("also", []):
    print(" also:", a.get("config_name")[:40], a.get("value"), a.get("ms_per_step"), a.get("ms_per_step_median"), a.get("error"), (a.get("cpu_baseline") or {}).get("value"))
P
