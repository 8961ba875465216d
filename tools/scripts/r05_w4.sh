#!/bin/bash
mkdir -p gpurun_out
{
echo "== wgrad check"; timeout 600 python tools/wino_wgrad_check.py --no-time 2>&1 | grep -v amdgpu.ids | tail -4
timeout 300 python tools/wino_time.py --what wgrad 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r05_w4.txt 2>&1
cat gpurun_out/r05_w4.txt
