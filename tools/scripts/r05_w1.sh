#!/bin/bash
mkdir -p gpurun_out
{
echo "== wgrad check (one-wave kernel)"; timeout 600 python tools/wino_wgrad_check.py --no-time 2>&1 | grep -v amdgpu.ids
echo "== time new"; timeout 300 python tools/wino_time.py --what wgrad --check 2>&1 | grep -v amdgpu.ids
echo "== time old"; TMF_WINO_P=0 timeout 300 python tools/wino_time.py --what wgrad 2>&1 | grep -v amdgpu.ids
echo "== pytest"; timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "wino or Wino" 2>&1 | tail -3
} > gpurun_out/r05_w1.txt 2>&1
cat gpurun_out/r05_w1.txt
