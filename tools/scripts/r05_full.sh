#!/bin/bash
# full GPU test suite + the default bench line (no `also`) -> gpurun_out/
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r05_gputest.txt
cat gpurun_out/r05_gputest.txt
timeout 600 python bench.py --no-also --no-cpu-baseline > gpurun_out/r05_bench_quick.json 2> gpurun_out/r05_bench_quick.err
tail -c 1500 gpurun_out/r05_bench_quick.json | head -c 1500; echo
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r05_bench_quick.json").read().strip().splitlines()[-1])
    print("VALUE", d["value"], d["ms_per_step"], d.get("ms_per_step_min"), d.get("ms_per_step_median"))
except Exception as e:
    print("bench parse failed", e)
PY
