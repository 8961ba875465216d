#!/bin/bash
mkdir -p gpurun_out
{
timeout 200 python tools/wino_time.py --what fwd --check 2>&1 | grep -v amdgpu.ids
for v in pabl4 pabl32 pabl64 pbs1 pbs0; do TMF_LIB=transmf_ad_amd/libtmf_$v.so timeout 200 python tools/wino_time.py --what fwd 2>&1 | grep -v amdgpu.ids; done
} > gpurun_out/r05_p7.txt 2>&1
cat gpurun_out/r05_p7.txt
