# One GPU-box pass that regenerates what profiles/ holds for the round (fp32 headline + configs[2] bf16 report).
# Run through gpurun from the repo root:  gpurun --timeout 2400 -- 'bash tools/scripts/refresh_profiles.sh <tag>'
# then copy gpurun_out/refresh/* into profiles/.
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/refresh
rm -rf $O; mkdir -p $O
cd $R
# --- bench lines (un-profiled) ---
timeout 600 python3 bench.py > $O/b.log 2>&1; grep "^{" $O/b.log | tail -1 > $O/${TAG}_bench_n1.json
TMF_BENCH_SETUP_STEPS=0 timeout 300 python3 bench.py --no-cpu-baseline > $O/b.log 2>&1; grep "^{" $O/b.log | tail -1 > $O/${TAG}_bench_n1_setup0.json
TMF_DDP_FORCE=1 timeout 300 python3 bench.py --no-cpu-baseline > $O/b.log 2>&1; grep "^{" $O/b.log | tail -1 > $O/${TAG}_ddp_1rank.json
TMF_FUSION_FUSED=0 timeout 300 python3 bench.py --no-cpu-baseline > $O/b.log 2>&1; grep "^{" $O/b.log | tail -1 > $O/${TAG}_bench_n1_fusion_per_op.json
timeout 300 python3 bench.py --no-item-sync --no-cpu-baseline > $O/b.log 2>&1; grep "^{" $O/b.log | tail -1 > $O/${TAG}_bench_n1_no_item_sync.json
timeout 300 python3 bench.py --from-host --no-cpu-baseline > $O/b.log 2>&1; grep "^{" $O/b.log | tail -1 > $O/${TAG}_bench_n1_from_host.json
timeout 600 python3 bench.py --precision bf16 --storage bf16 --size 128 > $O/b.log 2>&1; grep "^{" $O/b.log | tail -1 > $O/${TAG}_bench_128_bf16_storage.json
timeout 300 python3 bench.py --precision bf16 --size 128 --no-cpu-baseline > $O/b.log 2>&1; grep "^{" $O/b.log | tail -1 > $O/${TAG}_bench_128_bf16.json
timeout 300 python3 bench.py --precision bf16 --storage bf16 --no-cpu-baseline > $O/b.log 2>&1; grep "^{" $O/b.log | tail -1 > $O/${TAG}_bench_96_bf16_storage.json
timeout 300 python3 bench.py --size 128 --no-cpu-baseline > $O/b.log 2>&1; grep "^{" $O/b.log | tail -1 > $O/${TAG}_bench_128_fp32.json
timeout 300 python3 bench.py --shape 91 109 91 --no-cpu-baseline > $O/b.log 2>&1; grep "^{" $O/b.log | tail -1 > $O/${TAG}_bench_adni_shape.json
timeout 300 python3 bench.py --model cnn --batch 16 --no-cpu-baseline > $O/b.log 2>&1; grep "^{" $O/b.log | tail -1 > $O/${TAG}_bench_cnn_b16.json
timeout 300 python3 bench.py --model single --batch 16 --no-cpu-baseline > $O/b.log 2>&1; grep "^{" $O/b.log | tail -1 > $O/${TAG}_bench_single_b16.json
timeout 300 python3 bench.py --precision fp32x --no-cpu-baseline > $O/b.log 2>&1; grep "^{" $O/b.log | tail -1 > $O/${TAG}_bench_n1_fp32x.json
timeout 300 python3 bench.py --eval --no-cpu-baseline > $O/b.log 2>&1; grep "^{" $O/b.log | tail -1 > $O/${TAG}_bench_eval_n1.json
timeout 300 python3 bench.py --eval --precision bf16 --storage bf16 --size 128 --no-cpu-baseline > $O/b.log 2>&1; grep "^{" $O/b.log | tail -1 > $O/${TAG}_bench_eval_128_bf16_storage.json
timeout 300 python3 bench.py --dropout 0.3 --no-cpu-baseline > $O/b.log 2>&1; grep "^{" $O/b.log | tail -1 > $O/${TAG}_bench_n1_dropout03.json
TMF_DDP_FORCE=1 TMF_DDP_INPLACE=0 timeout 300 python3 bench.py --no-cpu-baseline > $O/b.log 2>&1; grep "^{" $O/b.log | tail -1 > $O/${TAG}_ddp_1rank_buckets.json
timeout 300 python3 bench.py --no-also --no-cpu-baseline --steps 40 > $O/b.log 2>&1; grep "^{" $O/b.log | tail -1 > $O/${TAG}_bench_n1_plain40.json
TMF_DDP_FORCE=1 timeout 300 python3 bench.py --no-cpu-baseline --steps 40 > $O/b.log 2>&1; grep "^{" $O/b.log | tail -1 > $O/${TAG}_ddp_1rank_40.json
# --- kernel-trace statistics of the SAME commands (roofline loop only: the averages the roofline object quotes) ---
prof() { # name, bench flags...
  n=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/_p -o p -- python3 bench.py "$@" > $O/_p.log 2>&1
  cp $O/_p/p_kernel_stats.csv $O/${TAG}_${n}_kernel_stats.csv; grep "^{" $O/_p.log | tail -1 > $O/${TAG}_${n}.json; rm -rf $O/_p
}
prof roofline_only --roofline-only --no-cpu-baseline
prof roofline_only_128_bf16_storage --roofline-only --no-cpu-baseline --precision bf16 --storage bf16 --size 128
# --- kernel-trace statistics of whole train steps (two encoder streams, and one stream for additive numbers) ---
TMF_ROOF_REPS=1 TMF_ROOF_SPIN_S=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/_p -o p -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/_p.log 2>&1; cp $O/_p/p_kernel_stats.csv $O/${TAG}_bench_kernel_stats.csv; rm -rf $O/_p
TMF_STREAMS=1 TMF_ROOF_REPS=1 TMF_ROOF_SPIN_S=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/_p -o p -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/_p.log 2>&1; cp $O/_p/p_kernel_stats.csv $O/${TAG}_bench_kernel_stats_1stream.csv; rm -rf $O/_p
TMF_STREAMS=1 TMF_ROOF_REPS=1 TMF_ROOF_SPIN_S=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/_p -o p -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --precision bf16 --storage bf16 --size 128 > $O/_p.log 2>&1; cp $O/_p/p_kernel_stats.csv $O/${TAG}_bench_128_bf16_storage_kernel_stats_1stream.csv; rm -rf $O/_p
# --- GPU busy / idle accounting of the two-stream steps (largest gaps with the kernels on either side) ---
TMF_ROOF_REPS=1 TMF_ROOF_SPIN_S=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/_p -o p -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/_p.log 2>&1; python3 tools/trace_gaps.py $O/_p/p_kernel_trace.csv --steps 8 --top 30 > $O/${TAG}_trace_gaps_fp32.txt 2>&1; rm -rf $O/_p
TMF_ROOF_REPS=1 TMF_ROOF_SPIN_S=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/_p -o p -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --precision bf16 --storage bf16 --size 128 > $O/_p.log 2>&1; python3 tools/trace_gaps.py $O/_p/p_kernel_trace.csv --steps 8 --top 30 > $O/${TAG}_trace_gaps_128_bf16_storage.txt 2>&1; rm -rf $O/_p
rm -f $O/b.log $O/_p.log
ls -la $O
