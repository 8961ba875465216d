cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04i
rm -rf $O; mkdir -p $O
cd $R
for i in 1 2; do
python3 bench.py --no-also --no-cpu-baseline --steps 40 > $O/plain$i.json 2>> $O/err.log
TMF_DDP_FORCE=1 python3 bench.py --no-cpu-baseline --steps 40 > $O/ddp$i.json 2>> $O/err.log
TMF_DDP_FORCE=1 TMF_DDP_NOSYNC=1 python3 bench.py --no-cpu-baseline --steps 40 > $O/nosync$i.json 2>> $O/err.log
TMF_DDP_FORCE=1 TMF_DDP_INPLACE=0 python3 bench.py --no-cpu-baseline --steps 40 > $O/buckets$i.json 2>> $O/err.log
done
python3 - <<P
import json
for n in ("plain1","ddp1","nosync1","buckets1","plain2","ddp2","nosync2","buckets2"):
    try:
        d=json.loads(open("$O/"+n+".json").read().strip().splitlines()[-1]); print(n, d["value"], d["ms_per_step"], d.get("ms_per_step_min"), d.get("ms_per_step_median"), (d.get("per_rank") or {}).get("allreduce_exposed_ms_mean"))
    except Exception as e: print(n, "ERR", e)
P
