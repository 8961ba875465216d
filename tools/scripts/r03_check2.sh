cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/c2
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_input_pipeline.py -x -q > $O/t_input.log 2>&1; echo "rc=$?" >> $O/t_input.log
tail -8 $O/t_input.log
timeout 600 python3 -m pytest tests/test_gpu_model.py -x -q -m gpu -k "adam or heads" > $O/t_misc.log 2>&1; echo "rc=$?" >> $O/t_misc.log
tail -5 $O/t_misc.log
# instrumented build (phase time stamps) + timeline of the fused kernels
export TMF_EXTRA_FLAGS=-DTMF_XF_TRACE
python3 -m transmf_ad_amd.build > $O/build.log 2>&1; tail -2 $O/build.log
timeout 300 python3 tools/xf_trace.py 216 8 > $O/trace_216.txt 2>&1; cat $O/trace_216.txt
timeout 300 python3 tools/xf_trace.py 512 8 > $O/trace_512.txt 2>&1; cat $O/trace_512.txt
