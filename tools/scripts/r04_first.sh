# what the driver's run sees: bench.py as the FIRST process on a fresh box, per-step times listed
TMF_BENCH_STEP_TIMES=1 python bench.py --no-also --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print(j['value'], j['ms_per_step'], j.get('ms_per_step_median')); print(j.get('ms_per_step_list'))
"
