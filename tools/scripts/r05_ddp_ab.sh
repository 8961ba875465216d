# A/B of the data-parallel machinery on one GPU: plain model vs GradAllReduce over a 1-rank RCCL group (TMF_DDP_FORCE=1; in place,
# and TMF_DDP_INPLACE=0 = everything through the end-of-backward buckets), alternating processes, 60 timed steps each.
#   gpurun -- 'bash tools/scripts/r05_ddp_ab.sh'  -> gpurun_out/r05ddp/ddp_overhead.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05ddp
rm -rf $O; mkdir -p $O
cd $R
for i in 1 2 3 4; do
python3 bench.py --no-also --no-cpu-baseline --steps 60 > $O/plain$i.json 2>> $O/err.log
TMF_DDP_FORCE=1 python3 bench.py --no-cpu-baseline --steps 60 > $O/ddp$i.json 2>> $O/err.log
TMF_DDP_FORCE=1 TMF_DDP_INPLACE=0 python3 bench.py --no-cpu-baseline --steps 60 > $O/buckets$i.json 2>> $O/err.log
done
python3 - <<P > $O/ddp_overhead.txt
import json, statistics
print("# tools/scripts/r05_ddp_ab.sh: bench.py --steps 60, alternating processes on one box; ms per step: wall mean / per-step min / median; exposed all-reduce (ms)")
rows = {}
for kind in ("plain", "ddp", "buckets"):
    for i in (1, 2, 3, 4):
        d = json.loads(open("$O/%s%d.json" % (kind, i)).read().strip().splitlines()[-1])
        rows.setdefault(kind, []).append(d)
        pr = d.get("per_rank") or {}
        print(f"{kind:8s} run {i}: {d['value']:8.1f} pairs/s  {d['ms_per_step']:7.3f} / {d['ms_per_step_min']:7.3f} / {d['ms_per_step_median']:7.3f}   exposed {pr.get('allreduce_exposed_ms_mean')}  {pr.get('collective_kinds')}")
med = {k: statistics.median(d["ms_per_step_median"] for d in v) for k, v in rows.items()}
mn = {k: min(d["ms_per_step_min"] for d in v) for k, v in rows.items()}
print(f"median of the per-step medians: plain {med['plain']:.3f} ms, in place {med['ddp']:.3f} (+{(med['ddp'] / med['plain'] - 1) * 100:.2f} %), buckets {med['buckets']:.3f} (+{(med['buckets'] / med['plain'] - 1) * 100:.2f} %)")
print(f"best step of any run:          plain {mn['plain']:.3f} ms, in place {mn['ddp']:.3f} (+{(mn['ddp'] / mn['plain'] - 1) * 100:.2f} %), buckets {mn['buckets']:.3f} (+{(mn['buckets'] / mn['plain'] - 1) * 100:.2f} %)")
P
cat $O/ddp_overhead.txt
