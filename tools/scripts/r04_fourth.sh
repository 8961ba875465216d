cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04g
rm -rf $O; mkdir -p $O
cd $R
timeout 3000 python3 -m pytest tests/ -q -m gpu > $O/t_all.log 2>&1; echo "rc=$?" >> $O/t_all.log
tail -8 $O/t_all.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
