#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
{
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "pair_sums or conv_bn_act_pool or first_block" 2>&1 | tail -5
for g in 1 0 1 0; do TMF_C1_GRAM=$g timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench gram=$g', d['value'], d['ms_per_step'], d.get('ms_per_step_min'), d.get('ms_per_step_median'))"; done
TMF_STREAMS=1 TMF_ROOF_REPS=1 TMF_ROOF_SPIN_S=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/_p -o p -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also > gpurun_out/_p.log 2>&1
grep -E "c1_|conv1_fused|bn_finalize" gpurun_out/_p/p_kernel_stats.csv | cut -c1-160
rm -rf gpurun_out/_p
} > gpurun_out/r05_g1.txt 2>&1
cat gpurun_out/r05_g1.txt
