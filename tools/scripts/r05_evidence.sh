#!/bin/bash
# round 5: the per-kernel evidence behind DESIGN 3.15 / 3.16 in one GPU-box pass -> gpurun_out/r05ev/ (copied to profiles/ by hand)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05ev; rm -rf $O; mkdir -p $O; cd $R
{ echo "# tools/microbench/mix_cost.hip: the instruction mix of one 8-channel chunk of the Winograd forward beside its MFMAs, one wave per SIMD (B) against two (D)"; ./tools/microbench/mix_cost; } > $O/r05_mix_cost.txt 2>&1
{ echo "# tools/wino_time.py: back-to-back launches at B = 8, 96^3 input (us; in brackets executed matrix flops / 157.3 TF)"
  echo "# persistent one-wave-per-SIMD kernels (default)"; timeout 300 python tools/wino_time.py --what fwd,dgrad,wgrad 2>&1 | grep -v amdgpu.ids
  echo "# two-waves-per-SIMD kernels of round 4 (TMF_WINO_P=0)"; TMF_WINO_P=0 timeout 300 python tools/wino_time.py --what fwd,dgrad,wgrad --tag wino_p=0 2>&1 | grep -v amdgpu.ids
  echo "# again, persistent"; timeout 300 python tools/wino_time.py --what fwd,dgrad,wgrad 2>&1 | grep -v amdgpu.ids
} > $O/r05_wino_time.txt 2>&1
{ echo "# tools/wino_check.py / tools/wino_wgrad_check.py: the Winograd entries against fp64 torch and the direct kernels"
  timeout 600 python tools/wino_check.py --no-time 2>&1 | grep -v amdgpu.ids; timeout 600 python tools/wino_wgrad_check.py --no-time 2>&1 | grep -v amdgpu.ids; } > $O/r05_wino_check.txt 2>&1
timeout 300 python tools/host_timeline.py 2>&1 | grep -v amdgpu.ids > $O/r05_host_timeline.txt
{ for g in 1 0 1 0; do TMF_C1_GRAM=$g timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('TMF_C1_GRAM=$g', d['value'], 'pairs/s', d['ms_per_step'], 'ms mean', d.get('ms_per_step_min'), 'min', d.get('ms_per_step_median'), 'median')"; done
  for s in 2 1 2 1; do TMF_STREAMS=$s timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('TMF_STREAMS=$s', d['value'], 'pairs/s', d['ms_per_step'], 'ms mean', d.get('ms_per_step_min'), 'min', d.get('ms_per_step_median'), 'median')"; done
} > $O/r05_step_ab.txt 2>&1
cat $O/r05_mix_cost.txt $O/r05_wino_time.txt $O/r05_step_ab.txt $O/r05_host_timeline.txt | cut -c1-260
