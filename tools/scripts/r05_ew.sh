#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
{
for l in hip ew2048 ew1024 hip ew2048 ew1024; do TMF_LIB=transmf_ad_amd/libtmf_$l.so timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$l', d['value'], d['ms_per_step'], d.get('ms_per_step_min'), d.get('ms_per_step_median'))"; done
for l in hip ew1024; do
TMF_LIB=transmf_ad_amd/libtmf_$l.so TMF_STREAMS=1 TMF_ROOF_REPS=1 TMF_ROOF_SPIN_S=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/_p -o p -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also > gpurun_out/_p.log 2>&1
echo "== $l"; grep -E "bn_" gpurun_out/_p/p_kernel_stats.csv | cut -d, -f1-4 | sed 's/(anonymous namespace):://; s/void //' | cut -c1-60,100-200
rm -rf gpurun_out/_p
done
} > gpurun_out/r05_ew.txt 2>&1
cat gpurun_out/r05_ew.txt
