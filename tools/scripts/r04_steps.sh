cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04h
rm -rf $O; mkdir -p $O
cd $R
for i in 1 2 3; do
TMF_BENCH_STEP_TIMES=1 python3 bench.py --no-also --no-cpu-baseline --steps 20 > $O/b$i.json 2> $O/b$i.err
python3 - <<P
import json
d=json.loads(open("$O/b$i.json").read().strip().splitlines()[-1])
print("run $i", d["value"], d["ms_per_step"], d["ms_per_step_min"], d["ms_per_step_median"], d["ms_per_step_list"])
P
done
