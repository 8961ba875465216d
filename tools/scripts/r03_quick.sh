cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/qk
rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests/test_gpu_kernels.py -q -m gpu -x > $O/t.log 2>&1; echo "rc=$?" >> $O/t.log
tail -5 $O/t.log
for i in 1 2; do
python3 bench.py --size 128 --precision bf16 --storage bf16 --steps 20 --warmup 5 --no-cpu-baseline > $O/b128_$i.json 2> $O/b128_$i.err; python3 - <<P
import json; d=json.loads(open("$O/b128_$i.json").read().strip().splitlines()[-1]); print("128bf16", d["value"], d["ms_per_step"], d["roofline"]["frac"])
P
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/b96_$i.json 2> $O/b96_$i.err; python3 - <<P
import json; d=json.loads(open("$O/b96_$i.json").read().strip().splitlines()[-1]); print("96fp32", d["value"], d["ms_per_step"], d["roofline"]["frac"])
P
done
