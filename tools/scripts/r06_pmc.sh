# round 6 counter passes (each its own process; --pmc only with --kernel-trace):
#  (1) HBM traffic per launch of every conv kernel instance of bench.py's roofline loop (FETCH_SIZE / WRITE_SIZE, separate passes),
#      fp32 96^3 (the contract line) and bf16-storage 128^3 (configs[2])
#  (2) the whole fp32 train step: matrix-pipe busy, clock and VALU instructions per MFMA per kernel family
#   gpurun -- 'bash tools/scripts/r06_pmc.sh'   -> gpurun_out/r06pmc/*
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06pmc
rm -rf $O; mkdir -p $O
cd $R
run() { n=$1; c=$2; shift 2
  TMF_ROOF_REPS=3 TMF_ROOF_SPIN_S=0.3 timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/$n -o $n --output-format csv -- python3 bench.py --roofline-only --no-cpu-baseline "$@" > $O/$n.log 2>&1
}
run f32_fetch FETCH_SIZE
run f32_write WRITE_SIZE
run b16_fetch FETCH_SIZE --precision bf16 --storage bf16 --size 128
run b16_write WRITE_SIZE --precision bf16 --storage bf16 --size 128
python3 tools/pmc_traffic.py $O/f32_fetch $O/f32_write $O/traffic_f32.json --note "fp32 (Winograd default: split kernel forward / data gradient, persistent fp32 weight gradient), B=8, 96^3" > $O/traffic_f32.txt
python3 tools/pmc_traffic.py $O/b16_fetch $O/b16_write $O/traffic_b16.json --note "bf16 storage, B=8, 128^3" > $O/traffic_b16.txt
python3 - <<PY
import json
a = json.load(open("$O/traffic_f32.json")); b = json.load(open("$O/traffic_b16.json"))
json.dump({"fp32|fp32|8|96x96x96": a, "bf16|bf16|8|128x128x128": b}, open("$O/r06_pmc_traffic.json", "w"), indent=1)
PY
TMF_BENCH_SETUP_STEPS=3 timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA -d $O/step -o step --output-format csv -- python3 $R/bench.py --steps 10 --warmup 2 --no-also --no-cpu-baseline > $O/step.log 2>&1
python3 - <<PY > $O/r06_pmc_step_busy_winograd.txt
import csv, glob, collections, re
per = collections.defaultdict(dict)
for f in glob.glob("$O/step/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        e = per[int(r["Dispatch_Id"])]
        e["name"] = r["Kernel_Name"]
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        e["us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
fam = collections.defaultdict(lambda: [0.0, 0.0, 0.0, 0, 0.0, 0.0])
def family(n):
    m = re.search(r"(conv3d_wino_wgrad_p_kernel<\d>|conv3d_wino_p_kernel<\d, \d>|conv3d_winox_kernel<\d>)", n)
    if m:
        return m.group(1)
    for key in ("conv3d_wino_wgrad_kernel", "conv3d_wino_kernel", "wino_", "conv3d_fwd_kernel", "conv3d_wgrad_kernel", "conv1_fused_kernel", "conv1x1", "bn_", "xf_", "tok_", "slab_reduce", "heads", "adam"):
        if key in n:
            return key
    return "other"
for e in per.values():
    f = fam[family(e["name"])]
    f[0] += e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0); f[1] += e.get("GRBM_GUI_ACTIVE", 0.0) / 8.0; f[2] += e["us"]; f[3] += 1
    f[4] += e.get("SQ_INSTS_VALU", 0.0); f[5] += e.get("SQ_INSTS_MFMA", 0.0)
tb = sum(f[0] for f in fam.values()); ta = sum(f[1] for f in fam.values()); tu = sum(f[2] for f in fam.values())
print("# tools/scripts/r06_pmc.sh (2): rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA -- python3 bench.py --steps 10 --warmup 2 --no-also --no-cpu-baseline (+ 3 set-up steps, the roofline loop's launches)")
print("# every kernel of the process, serialised by the counter collection; busy = MFMA busy cycles / (1024 SIMDs x active cycles)")
for k, f in sorted(fam.items(), key=lambda kv: -kv[1][2]):
    print(f"{k:32s} launches {f[3]:6d}  time {f[2] / 1e3:9.2f} ms  clock {f[1] / max(f[2], 1e-9) / 1e3:5.2f} GHz  matrix pipe busy {f[0] / max(1024.0 * f[1], 1e-9):6.3f}  VALU per MFMA {f[4] / max(f[5], 1.0):6.2f}")
print(f"{'ALL':32s} time {tu / 1e3:9.2f} ms  matrix pipe busy {tb / (1024.0 * ta):6.3f} of the active cycles (64 busy cycles per v_mfma_f32_32x32x2_f32, 32 per v_mfma_f32_32x32x16_bf16)")
PY
cat $O/r06_pmc_step_busy_winograd.txt; tail -2 $O/step.log; cat $O/traffic_f32.txt
rm -rf $O/step $O/f32_fetch $O/f32_write $O/b16_fetch $O/b16_write
