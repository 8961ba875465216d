#!/bin/bash
mkdir -p gpurun_out
{
timeout 300 python tools/wino_time.py --what wgrad 2>&1 | grep -v amdgpu.ids
for v in 1 4 5 16; do TMF_LIB=transmf_ad_amd/libtmf_wabl$v.so timeout 300 python tools/wino_time.py --what wgrad 2>&1 | grep -v amdgpu.ids; done
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_w2 -o w2 -- python tools/wino_time.py --what wgrad --rounds 1 --reps 10 > /dev/null 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/prof_w2/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("wgrad", "slab", "finish")):
            print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY
} > gpurun_out/r05_w2.txt 2>&1
cat gpurun_out/r05_w2.txt
