#!/bin/bash
# Timing ablations of the persistent Winograd kernels (profiles/r05_wino_ablations.txt).
#   on the build box:  bash tools/scripts/r05_wino_ablations.sh build      (variant libraries transmf_ad_amd/libtmf_{p,w}ablN.so)
#   then:              gpurun --timeout 1200 -- 'bash tools/scripts/r05_wino_ablations.sh'
# P_ABL bits (csrc/conv3d_wino.hip; results are wrong on purpose): 1 no input transform, 2 no weight loads, 4 no halo / stage copies,
# 8 no epilogue (forward), 16 no MFMAs, 128 no statistic sums, 256 no store_guard.
if [ "$1" = build ]; then
  for v in 1 2 4 8 16 15 31; do python tools/build_variant.py pabl$v --replace conv3d_wino.hip=transmf_ad_amd/csrc/conv3d_wino.hip --flags=-DP_ABL=$v; done
  for v in 1 4 5 16; do python tools/build_variant.py wabl$v --replace conv3d_wino.hip=transmf_ad_amd/csrc/conv3d_wino.hip --flags=-DP_ABL=$v; done
  exit 0
fi
mkdir -p gpurun_out
{
timeout 200 python tools/wino_time.py --what fwd,wgrad 2>&1 | grep -v amdgpu.ids
for v in 1 2 4 8 16 15 31; do TMF_LIB=transmf_ad_amd/libtmf_pabl$v.so timeout 200 python tools/wino_time.py --what fwd 2>&1 | grep -v amdgpu.ids; done
for v in 1 4 5 16; do TMF_LIB=transmf_ad_amd/libtmf_wabl$v.so timeout 300 python tools/wino_time.py --what wgrad 2>&1 | grep -v amdgpu.ids; done
} > gpurun_out/r05_wino_ablations.txt 2>&1
cat gpurun_out/r05_wino_ablations.txt
