#!/bin/bash
mkdir -p gpurun_out
{
./tools/microbench/valu_cost
echo "== trace"; TMF_LIB=transmf_ad_amd/libtmf_ptrace.so timeout 200 python tools/wino_ptrace.py 2>&1 | grep -v amdgpu.ids | head -30
echo "== time"; timeout 200 python tools/wino_time.py --what fwd,dgrad --check 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r05_p3.txt 2>&1
cat gpurun_out/r05_p3.txt
