#!/bin/bash
mkdir -p gpurun_out
{
echo "== trace (halo copies at the barrier)"; TMF_LIB=transmf_ad_amd/libtmf_ptrace.so timeout 200 python tools/wino_ptrace.py 2>&1 | grep -v amdgpu.ids
echo "== trace (halo copies behind the weights)"; TMF_LIB=transmf_ad_amd/libtmf_ptrace_late.so timeout 200 python tools/wino_ptrace.py 2>&1 | grep -v amdgpu.ids
echo "== trace conv3.3"; TMF_LIB=transmf_ad_amd/libtmf_ptrace_late.so timeout 200 python tools/wino_ptrace.py --cin 64 --cout 128 --S 24 2>&1 | grep -v amdgpu.ids
echo "== time"; timeout 200 python tools/wino_time.py --what fwd,dgrad 2>&1 | grep -v amdgpu.ids
TMF_LIB=transmf_ad_amd/libtmf_p_late.so timeout 200 python tools/wino_time.py --what fwd,dgrad --check 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r05_p2.txt 2>&1
cat gpurun_out/r05_p2.txt
