#!/bin/bash
mkdir -p gpurun_out
{
echo "== wgrad check"; timeout 600 python tools/wino_wgrad_check.py --no-time 2>&1 | grep -v amdgpu.ids | tail -3
echo "== fwd check"; timeout 600 python tools/wino_check.py --no-time 2>&1 | grep -v amdgpu.ids | tail -3
timeout 300 python tools/wino_time.py --what fwd,dgrad,wgrad 2>&1 | grep -v amdgpu.ids
echo "== pytest"; timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "wino or Wino" 2>&1 | tail -3
} > gpurun_out/r05_w5.txt 2>&1
cat gpurun_out/r05_w5.txt
