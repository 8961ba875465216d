#!/bin/bash
# parity margins with the first-block statistics from pair sums (TMF_C1_GRAM=1, default) and from the recomputing pass (0)
mkdir -p gpurun_out
{
for g in 1 0; do
  echo "== TMF_C1_GRAM=$g"
  TMF_C1_GRAM=$g timeout 900 python tools/parity_report.py --cases ad_mid,ad_full_b2,ad_full_b2_blobs,ad_adni_b2,ad_full_b8 --modes fp32,fp32x 2>&1 | grep -v amdgpu.ids
done
} > gpurun_out/r05_parity_gram.txt 2>&1
cat gpurun_out/r05_parity_gram.txt
