#!/bin/bash
mkdir -p gpurun_out
{
echo "== trace"; TMF_LIB=transmf_ad_amd/libtmf_ptrace.so timeout 200 python tools/wino_ptrace.py 2>&1 | grep -v amdgpu.ids | head -26
echo "== check"; timeout 300 python tools/wino_check.py --no-time 2>&1 | grep -v amdgpu.ids
echo "== time"; timeout 200 python tools/wino_time.py --what fwd,dgrad 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r05_p4.txt 2>&1
cat gpurun_out/r05_p4.txt
