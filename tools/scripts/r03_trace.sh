# kernel trace of the two-stream fp32 step: GPU busy / idle accounting + per-kernel table; same for 128^3 bf16 storage
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/tr
rm -rf $O; mkdir -p $O
cd $R
TMF_ROOF_REPS=1 TMF_ROOF_SPIN_S=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/_p -o p -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/_p.log 2>&1
python3 tools/trace_gaps.py $O/_p/p_kernel_trace.csv --steps 8 --top 12 > $O/gaps_fp32.txt 2>&1; cat $O/gaps_fp32.txt
rm -rf $O/_p
TMF_ROOF_REPS=1 TMF_ROOF_SPIN_S=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/_p -o p -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --precision bf16 --storage bf16 --size 128 > $O/_p.log 2>&1
python3 tools/trace_gaps.py $O/_p/p_kernel_trace.csv --steps 8 --top 12 > $O/gaps_bf16.txt 2>&1; cat $O/gaps_bf16.txt
rm -rf $O/_p
