cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c3
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "fused_fusion" > $O/t_kernels.log 2>&1; echo "rc=$?" >> $O/t_kernels.log
tail -5 $O/t_kernels.log
timeout 900 python3 -m pytest tests/test_input_pipeline.py -x -q > $O/t_input.log 2>&1; echo "rc=$?" >> $O/t_input.log
tail -4 $O/t_input.log
timeout 1500 python3 -m pytest tests/test_gpu_model.py -x -q -m gpu -k "golden or one_call_fusion or full_size or 128_cubed" > $O/t_model.log 2>&1; echo "rc=$?" >> $O/t_model.log
tail -4 $O/t_model.log
timeout 300 python3 bench.py --no-cpu-baseline > $O/b.log 2>&1; grep "^{" $O/b.log | tail -1 > $O/bench_n1.json
python3 -c "import json;d=json.load(open('$O/bench_n1.json'));print(d['value'],d['ms_per_step'])"
timeout 300 python3 bench.py --precision bf16 --storage bf16 --size 128 --no-cpu-baseline > $O/b.log 2>&1; grep "^{" $O/b.log | tail -1 > $O/bench_128_bf16s.json
python3 -c "import json;d=json.load(open('$O/bench_128_bf16s.json'));print(d['value'],d['ms_per_step'])"
export TMF_EXTRA_FLAGS=-DTMF_XF_TRACE
python3 -m transmf_ad_amd.build > $O/build.log 2>&1; tail -1 $O/build.log
timeout 300 python3 tools/xf_trace.py 216 8 > $O/trace_216.txt 2>&1; cat $O/trace_216.txt
timeout 300 python3 tools/xf_trace.py 512 8 > $O/trace_512.txt 2>&1; cat $O/trace_512.txt
