# HBM traffic (PMC) of the conv kernel instances as bench.py's roofline loop launches them: fp32 96^3 and bf16-storage 128^3.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmctraffic
rm -rf $O; mkdir -p $O
cd $R
run() { n=$1; c=$2; shift 2
  TMF_ROOF_REPS=3 TMF_ROOF_SPIN_S=0.3 timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/$n -o $n --output-format csv -- python3 bench.py --roofline-only --no-cpu-baseline "$@" > $O/$n.log 2>&1
}
run f32_fetch FETCH_SIZE
run f32_write WRITE_SIZE
run b16_fetch FETCH_SIZE --precision bf16 --storage bf16 --size 128
run b16_write WRITE_SIZE --precision bf16 --storage bf16 --size 128
python3 tools/pmc_traffic.py $O/f32_fetch $O/f32_write $O/traffic_f32.json --note "fp32, B=8, 96^3"
python3 tools/pmc_traffic.py $O/b16_fetch $O/b16_write $O/traffic_b16.json --note "bf16 storage, B=8, 128^3"
python3 - <<PY
import json
a = json.load(open("$O/traffic_f32.json")); b = json.load(open("$O/traffic_b16.json"))
json.dump({"fp32|fp32|8|96x96x96": a, "bf16|bf16|8|128x128x128": b}, open("$O/r04_pmc_traffic.json", "w"), indent=1)
PY
