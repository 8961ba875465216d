cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/bf
rm -rf $O; mkdir -p $O
cd $R
timeout 1200 python3 -m pytest tests/test_gpu_kernels.py tests/test_kernel_resources.py -q -m gpu -x -k "bf16" > $O/t.log 2>&1; echo "rc=$?" >> $O/t.log
tail -8 $O/t.log
python3 tools/bf16_ab.py --dbg 0 2>&1 | grep -v amdgpu.ids
for i in 1 2; do
python3 bench.py --size 128 --precision bf16 --storage bf16 --steps 20 --warmup 5 --no-cpu-baseline > $O/b128_$i.json 2> $O/b128_$i.err; python3 - <<P
import json; d=json.loads(open("$O/b128_$i.json").read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("achieved"))
P
done
