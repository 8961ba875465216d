cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/qk
rm -rf $O; mkdir -p $O
cd $R
timeout 2500 python3 -m pytest tests/ -q -m gpu > $O/t.log 2>&1; echo "rc=$?" >> $O/t.log
tail -8 $O/t.log
for i in 1 2 3; do
TMF_CONV_RT=0 python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline > $O/b96_rt0_$i.json 2> $O/b.err; python3 - <<P
import json; d=json.loads(open("$O/b96_rt0_$i.json").read().strip().splitlines()[-1]); print("96fp32 rt0", d["value"], d["ms_per_step"])
P
python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline > $O/b96_$i.json 2> $O/b.err; python3 - <<P
import json; d=json.loads(open("$O/b96_$i.json").read().strip().splitlines()[-1]); print("96fp32 rt1", d["value"], d["ms_per_step"])
P
done
