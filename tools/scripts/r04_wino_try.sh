#!/bin/bash
# first look at the Winograd form in the whole step: the GPU suite with the mode forced on, then the bench line for each mode
mkdir -p gpurun_out/wino
TMF_CONV_WINO=2 timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_kernels.py 2>&1 | tail -15 > gpurun_out/wino/pytest_wino2.txt
for m in 0 2 1 0 2; do
  TMF_CONV_WINO=$m timeout 300 python bench.py --no-also --no-cpu-baseline --steps 40 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('wino$m', j['value'], j['ms_per_step'], j.get('ms_per_step_median'), j.get('numerics_gate'))
" >> gpurun_out/wino/bench.txt
done
cat gpurun_out/wino/pytest_wino2.txt gpurun_out/wino/bench.txt
