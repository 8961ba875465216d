cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04c
rm -rf $O; mkdir -p $O
cd $R
timeout 1200 python3 -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "bf16" > $O/t_bf16.log 2>&1; echo "rc=$?" >> $O/t_bf16.log
tail -6 $O/t_bf16.log
timeout 600 python3 -m pytest tests/test_gpu_parallel.py -q -m gpu > $O/t_par.log 2>&1; echo "rc=$?" >> $O/t_par.log
tail -6 $O/t_par.log
python3 tools/bf16_ab.py --opt bf16_r5 --dbg 0,1 --only conv2 > $O/ab_r5.txt 2>&1; cat $O/ab_r5.txt
