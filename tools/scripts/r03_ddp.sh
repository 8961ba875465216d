cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ddp
rm -rf $O; mkdir -p $O
cd $R
timeout 300 python3 bench.py --no-cpu-baseline > $O/b.log 2>&1; grep "^{" $O/b.log | tail -1 > $O/plain.json
TMF_DDP_FORCE=1 timeout 300 python3 bench.py --no-cpu-baseline > $O/b.log 2>&1; grep "^{" $O/b.log | tail -1 > $O/ddp.json
python3 -c "
import json
a=json.load(open('$O/plain.json')); b=json.load(open('$O/ddp.json'))
print('plain', a['value'], a['ms_per_step'], 'ddp-1rank', b['value'], b['ms_per_step'], b.get('per_rank'))"
export TMF_DDP_FORCE=1
TMF_ROOF_REPS=1 TMF_ROOF_SPIN_S=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/_p -o p -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/_p.log 2>&1
python3 tools/trace_gaps.py $O/_p/p_kernel_trace.csv --steps 8 --top 40 > $O/gaps_ddp.txt 2>&1; cat $O/gaps_ddp.txt
rm -rf $O/_p
