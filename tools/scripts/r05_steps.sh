for i in 1 2 3; do
TMF_BENCH_STEP_TIMES=1 timeout 300 python bench.py --no-also --no-cpu-baseline > gpurun_out/_b.json 2>/dev/null
python - <<PY
import json
d = json.loads(open("gpurun_out/_b.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("ms_per_step_min"), d.get("ms_per_step_median"), d.get("ms_per_step_list"))
PY
done
