cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c4
rm -rf $O; mkdir -p $O
cd $R
timeout 3000 python3 -m pytest tests/ -x -q -m gpu > $O/t_all.log 2>&1; echo "rc=$?" >> $O/t_all.log
tail -4 $O/t_all.log
for i in 1 2; do
timeout 300 python3 bench.py --no-cpu-baseline > $O/b.log 2>&1; grep "^{" $O/b.log | tail -1 > $O/bench_n1_$i.json
python3 -c "import json;d=json.load(open('$O/bench_n1_$i.json'));print(d['value'],d['ms_per_step'])"
done
timeout 300 python3 bench.py --precision bf16 --storage bf16 --size 128 --no-cpu-baseline > $O/b.log 2>&1; grep "^{" $O/b.log | tail -1 > $O/bench_128_bf16s.json
python3 -c "import json;d=json.load(open('$O/bench_128_bf16s.json'));print(d['value'],d['ms_per_step'])"
TMF_ROOF_REPS=1 TMF_ROOF_SPIN_S=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/_p -o p -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/_p.log 2>&1
python3 tools/trace_gaps.py $O/_p/p_kernel_trace.csv --steps 8 --top 6 > $O/gaps_fp32.txt 2>&1; head -20 $O/gaps_fp32.txt
rm -rf $O/_p
