# Which kernels are scalar-issue bound?  SALU vs VALU instruction counts and scalar-unit activity per kernel over a few
# training steps (fp32 96^3 and bf16-storage 128^3).
#   gpurun -- 'bash tools/scripts/pmc_salu.sh'   -> gpurun_out/pmcsalu/summary_*.txt
cd /tmp && export TMPDIR=/tmp
export TMF_BENCH_SETUP_STEPS=2 TMF_ROOF_REPS=1 TMF_ROOF_SPIN_S=0
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmcsalu
rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/f32 -o f32 --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/f32.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/b16 -o b16 --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --precision bf16 --storage bf16 --size 128 > $O/b16.log 2>&1
cd $R
for n in f32 b16; do python3 - $O/$n <<'PY' > $O/summary_$n.txt
import csv, glob, sys, collections, re
per = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:70]
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_INSTS_SALU":
            cnt[k] += 1
rows = []
for k, c in per.items():
    gui = c["GRBM_GUI_ACTIVE"] / 8
    rows.append((gui, k, cnt[k], c["SQ_INSTS_SALU"] / max(c["SQ_INSTS_VALU"], 1), 4 * c["SQ_ACTIVE_INST_SCA"] / 1024 / max(gui, 1),
                 4 * c["SQ_ACTIVE_INST_VALU"] / 1024 / max(gui, 1)))
print(f"{'kernel':70s} {'calls':>5s} {'Mcycles':>8s} {'salu/valu':>9s} {'sca busy':>8s} {'valu busy':>9s}")
for gui, k, n, ratio, sca, valu in sorted(rows, reverse=True)[:40]:
    print(f"{k:70s} {n:5d} {gui / 1e6:8.3f} {ratio:9.2f} {sca:8.2f} {valu:9.2f}")
PY
done
cat $O/summary_f32.txt $O/summary_b16.txt
