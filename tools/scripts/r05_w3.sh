#!/bin/bash
mkdir -p gpurun_out
{
echo "== wgrad check"; timeout 600 python tools/wino_wgrad_check.py --no-time 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/wino_time.py --what wgrad 2>&1 | grep -v amdgpu.ids
for v in 1 4; do TMF_LIB=transmf_ad_amd/libtmf_wabl$v.so timeout 300 python tools/wino_time.py --what wgrad 2>&1 | grep -v amdgpu.ids; done
} > gpurun_out/r05_w3.txt 2>&1
cat gpurun_out/r05_w3.txt
