mkdir -p gpurun_out
for s in 2 1 2 1; do
  TMF_STREAMS=$s timeout 300 python bench.py --no-also --no-cpu-baseline > gpurun_out/_b.json 2>/dev/null
  python - <<PY
import json
d = json.loads(open("gpurun_out/_b.json").read().strip().splitlines()[-1])
print("streams $s:", d["value"], d["ms_per_step"], d.get("ms_per_step_min"), d.get("ms_per_step_median"))
PY
done
