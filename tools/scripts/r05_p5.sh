#!/bin/bash
mkdir -p gpurun_out
{
echo "== check"; timeout 300 python tools/wino_check.py --no-time 2>&1 | grep -v amdgpu.ids
echo "== pytest wino"; timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "wino or Wino" 2>&1 | tail -5
echo "== trace"; TMF_LIB=transmf_ad_amd/libtmf_ptrace.so timeout 200 python tools/wino_ptrace.py 2>&1 | grep -v amdgpu.ids | head -26
echo "== time"; timeout 200 python tools/wino_time.py --what fwd,dgrad 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r05_p5.txt 2>&1
cat gpurun_out/r05_p5.txt
