cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/hd
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_model.py -q -m gpu -x -k "heads or golden or property" > $O/t.log 2>&1; echo "rc=$?" >> $O/t.log
tail -25 $O/t.log
