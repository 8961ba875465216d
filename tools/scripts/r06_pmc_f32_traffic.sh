# HBM traffic per launch of the fp32 conv kernels only (the first half of r06_pmc.sh (1)); [TMF_LIB=...] bash tools/scripts/r06_pmc_f32_traffic.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06pmc_$1
rm -rf $O; mkdir -p $O
cd $R
run() { n=$1; c=$2; shift 2
  TMF_ROOF_REPS=3 TMF_ROOF_SPIN_S=0.3 timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/$n -o $n --output-format csv -- python3 bench.py --roofline-only --no-cpu-baseline "$@" > $O/$n.log 2>&1
}
run f32_fetch FETCH_SIZE
run f32_write WRITE_SIZE
python3 tools/pmc_traffic.py $O/f32_fetch $O/f32_write $O/traffic_f32.json --note "fp32, B=8, 96^3 ($1)" > $O/traffic_f32.txt
cat $O/traffic_f32.txt
rm -rf $O/f32_fetch $O/f32_write
