#!/bin/bash
mkdir -p gpurun_out
{
timeout 200 python tools/wino_time.py --what fwd --check 2>&1 | grep -v amdgpu.ids
for v in 1 2 4 8 16 15 31; do TMF_LIB=transmf_ad_amd/libtmf_pabl$v.so timeout 200 python tools/wino_time.py --what fwd 2>&1 | grep -v amdgpu.ids; done
} > gpurun_out/r05_p6.txt 2>&1
cat gpurun_out/r05_p6.txt
