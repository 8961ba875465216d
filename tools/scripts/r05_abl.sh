#!/bin/bash
# round 5, first GPU call: instruction-cost microbenchmark + ablation builds of the Winograd forward kernel
mkdir -p gpurun_out
./tools/microbench/valu_cost > gpurun_out/r05_valu_cost.txt 2>&1
python tools/wino_time.py --check > gpurun_out/r05_wino_abl.txt 2>&1
for v in 0 1 2 4 8 16 32 6 14 30 62; do
  TMF_LIB=transmf_ad_amd/libtmf_abl$v.so python tools/wino_time.py >> gpurun_out/r05_wino_abl.txt 2>&1
done
cat gpurun_out/r05_wino_abl.txt
tail -5 gpurun_out/r05_valu_cost.txt
