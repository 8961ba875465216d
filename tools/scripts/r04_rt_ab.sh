cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04rt
rm -rf $O; mkdir -p $O
cd $R
for i in 1 2 3; do
TMF_CONV_RT=0 python3 bench.py --no-also --no-cpu-baseline --steps 60 > $O/rt0_$i.json 2>> $O/err.log
TMF_CONV_RT=1 python3 bench.py --no-also --no-cpu-baseline --steps 60 > $O/rt1_$i.json 2>> $O/err.log
TMF_CONV_RT=2 python3 bench.py --no-also --no-cpu-baseline --steps 60 > $O/rt2_$i.json 2>> $O/err.log
done
python3 - <<P
import json
for n in ("rt0","rt1","rt2"):
    for i in (1,2,3):
        d=json.loads(open("$O/%s_%d.json"%(n,i)).read().strip().splitlines()[-1]); print(n, i, d["value"], d["ms_per_step"], d["ms_per_step_min"], d["ms_per_step_median"], d["roofline"]["step_conv"]["mfma_frac"])
P
