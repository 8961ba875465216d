#!/bin/bash
mkdir -p gpurun_out
timeout 900 python bench.py --no-also --no-cpu-baseline $BENCH_ARGS > gpurun_out/r05_bench_q.json 2> gpurun_out/r05_bench_q.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_bench_q.json").read().strip().splitlines()[-1])
print("VALUE", d["value"], d["ms_per_step"], d.get("ms_per_step_min"), d.get("ms_per_step_median"))
r = d["roofline"]
print("roofline", r["kernel"], r["frac"], "whole", r.get("whole_step", {}).get("mfma_frac"))
for k, v in r["kernels"].items(): print("  ", k[:40], v["ms"], v["mfma_frac"])
PY
