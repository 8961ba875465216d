cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04m
rm -rf $O; mkdir -p $O
cd $R
TMF_DDP_PROFILE=1 TMF_DDP_FORCE=1 python3 bench.py --no-cpu-baseline --steps 40 > $O/ddp.json 2> $O/err.log
grep "ddp profile" $O/ddp.json $O/err.log
