# first hardware pass over the fused fusion kernels: tests, bench line, per-kernel stats
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c1
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "fused_fusion" > $O/t_kernels.log 2>&1; echo "rc=$?" >> $O/t_kernels.log
tail -30 $O/t_kernels.log
timeout 1500 python3 -m pytest tests/test_gpu_model.py -x -q -m gpu -k "golden or one_call_fusion or full_size or 128_cubed or config3" > $O/t_model.log 2>&1; echo "rc=$?" >> $O/t_model.log
tail -15 $O/t_model.log
timeout 300 python3 bench.py --no-cpu-baseline > $O/b.log 2>&1; grep "^{" $O/b.log | tail -1 > $O/bench_n1.json
python3 -c "import json;d=json.load(open('$O/bench_n1.json'));print(d['value'],d['ms_per_step'])"
TMF_FUSION_FUSED=0 timeout 300 python3 bench.py --no-cpu-baseline > $O/b.log 2>&1; grep "^{" $O/b.log | tail -1 > $O/bench_n1_perop.json
python3 -c "import json;d=json.load(open('$O/bench_n1_perop.json'));print(d['value'],d['ms_per_step'])"
timeout 300 python3 bench.py --precision bf16 --storage bf16 --size 128 --no-cpu-baseline > $O/b.log 2>&1; grep "^{" $O/b.log | tail -1 > $O/bench_128_bf16s.json
python3 -c "import json;d=json.load(open('$O/bench_128_bf16s.json'));print(d['value'],d['ms_per_step'])"
TMF_STREAMS=1 TMF_ROOF_REPS=1 TMF_ROOF_SPIN_S=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/_p -o p -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/_p.log 2>&1; cp $O/_p/p_kernel_stats.csv $O/kernel_stats_1stream.csv; rm -rf $O/_p
TMF_STREAMS=1 TMF_ROOF_REPS=1 TMF_ROOF_SPIN_S=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/_p -o p -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --precision bf16 --storage bf16 --size 128 > $O/_p.log 2>&1; cp $O/_p/p_kernel_stats.csv $O/kernel_stats_128_bf16s_1stream.csv; rm -rf $O/_p
grep -i "xf_\|tok_wgrad\|heads\|token_pool" $O/kernel_stats_1stream.csv | cut -c1-160
