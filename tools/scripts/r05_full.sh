#!/bin/bash
# full GPU test suite + the default bench line -> gpurun_out/
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r05_gputest.txt
cat gpurun_out/r05_gputest.txt
timeout 900 python bench.py $BENCH_ARGS > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err
tail -5 gpurun_out/r05_bench.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r05_bench.json").read().strip().splitlines()[-1])
    print("VALUE", d["value"], d["ms_per_step"], d.get("ms_per_step_min"), d.get("ms_per_step_median"))
    r = d["roofline"]
    print("roofline", r["kernel"], r["frac"], r["achieved"], "alg", r.get("algorithmic_frac"), "whole", r.get("whole_step", {}).get("mfma_frac"))
    print("gate", json.dumps(d.get("numerics_gate"))[:900])
    print("cpu", d.get("cpu_baseline"))
    for a in d.get("also", []):
        print("also", a.get("config_name", "")[:50], a.get("value"), a.get("ms_per_step"), a.get("error"), (a.get("roofline") or {}).get("frac"))
except Exception as e:
    print("bench parse failed", e)
PY
