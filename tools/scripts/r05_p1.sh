#!/bin/bash
# round 5: first run of the persistent one-wave-per-SIMD Winograd forward kernel
mkdir -p gpurun_out
{
echo "== wino_check (default: persistent kernel)"; timeout 300 python tools/wino_check.py --no-time 2>&1 | grep -v amdgpu.ids
echo "== time: persistent"; timeout 200 python tools/wino_time.py --what fwd,dgrad --check 2>&1 | grep -v amdgpu.ids
echo "== time: two-waves kernel"; TMF_WINO_P=0 timeout 200 python tools/wino_time.py --what fwd,dgrad 2>&1 | grep -v amdgpu.ids
echo "== pytest wino"; timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "wino or Wino" 2>&1 | tail -5
} > gpurun_out/r05_p1.txt 2>&1
cat gpurun_out/r05_p1.txt
