# One GPU-box pass that regenerates what profiles/ holds for the fp32 headline configuration.
# Run through gpurun from the repo root: gpurun --timeout 1500 -- 'bash tools/scripts/refresh_profiles.sh'; then copy
# gpurun_out/refresh/* into profiles/ under the round's names.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/refresh
mkdir -p $O
cd $R
timeout 400 python3 bench.py > $O/bench_n1.log 2>&1; grep "^{" $O/bench_n1.log | tail -1 > $O/bench_n1.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/roof -o roof -- python3 bench.py --roofline-only > $O/roof.log 2>&1
grep "^{" $O/roof.log | tail -1 > $O/roofline_only.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b2s -o b2s -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/b2s.log 2>&1
TMF_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b1s -o b1s -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/b1s.log 2>&1
cp $O/roof/roof_kernel_stats.csv $O/roofline_only_kernel_stats.csv
cp $O/b2s/b2s_kernel_stats.csv $O/bench_kernel_stats.csv
cp $O/b1s/b1s_kernel_stats.csv $O/bench_kernel_stats_1stream.csv
rm -rf $O/roof $O/b2s $O/b1s
bash tools/scripts/pmc_fp32_hot.sh
cp gpurun_out/pmc32_summary.txt $O/pmc_conv2.3_hot.txt
python3 -c "
import json
d = json.load(open('$O/bench_n1.json')); r = json.load(open('$O/roofline_only.json'))
print('bench', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'])
print('roofline-only', r.get('roofline', r))
"
grep conv3d_fwd_kernel $O/roofline_only_kernel_stats.csv | head -3 | cut -c1-260
