# round 4 counter passes (each its own process; --pmc only with --kernel-trace):
#  (1) the bf16 forward kernel on random and on zero operands: effective clock (GRBM_GUI_ACTIVE / 8 XCDs / wall time) and
#      matrix-pipe busy cycles — what the chip's power budget takes
#  (2) the whole fp32 train step of bench.py (kernels serialised by the counter collection): matrix-pipe busy over all
#      kernels of a step = the headroom of the contract line as a counter (verdict r03 item 3)
#   gpurun -- 'bash tools/scripts/r04_pmc.sh'   -> gpurun_out/r04pmc/*.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04pmc
rm -rf $O; mkdir -p $O
run() { n=$1; what=$2; layer=$3; S=$4; fill=$5
  timeout 180 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA -d $O/$n -o $n --output-format csv -- python3 $R/tools/kone.py $what $layer --S $S --reps 8 --spin 60 --fill $fill > $O/$n.log 2>&1
}
run b16_conv2.0_randn bf16s conv2.0 128 randn
run b16_conv2.0_zeros bf16s conv2.0 128 zeros
run b16_conv2.3_randn bf16s conv2.3 128 randn
run b16_conv2.3_zeros bf16s conv2.3 128 zeros
run f32_conv2.3_randn fwd conv2.3 96 randn
run f32_conv2.3_zeros fwd conv2.3 96 zeros
cd $R
python3 - <<PY > $O/clock_busy.txt
import csv, glob, os, collections
print("# tools/scripts/r04_pmc.sh (1): per launch — wall time, effective clock = GRBM_GUI_ACTIVE / 8 / time, matrix-pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x active cycles), instructions")
for d in sorted(glob.glob("$O/*/")):
    n = os.path.basename(d.rstrip("/"))
    per = collections.defaultdict(dict)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "conv3d" not in k:
                continue
            e = per[int(r["Dispatch_Id"])]
            e["name"] = k.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").rsplit(">(", 1)[0] + ">"
            e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            e["us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    ds = sorted(per)
    if not ds:
        continue
    last = per[ds[-1]]["name"]
    sel = [per[i] for i in ds if per[i]["name"] == last][1:]
    if not sel:
        continue
    avg = lambda key: sum(e.get(key, 0.0) for e in sel) / len(sel)
    act = avg("GRBM_GUI_ACTIVE") / 8.0
    print(f"{n:20s} {last[:60]:60s} {avg('us'):8.1f} us  clock {act / avg('us') / 1e3:5.2f} GHz  pipe busy {avg('SQ_VALU_MFMA_BUSY_CYCLES') / (1024.0 * act):5.3f}"
          f"  wave quad-cycles {avg('SQ_WAVE_CYCLES') / 1e6:7.1f} M  VALU {avg('SQ_INSTS_VALU') / 1e6:6.2f} M  MFMA {avg('SQ_INSTS_MFMA') / 1e6:6.2f} M")
PY
cat $O/clock_busy.txt
# (2) whole step
TMF_BENCH_SETUP_STEPS=3 timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/step -o step --output-format csv -- python3 $R/bench.py --steps 10 --warmup 2 --no-also --no-cpu-baseline > $O/step.log 2>&1
python3 - <<PY > $O/step_busy.txt
import csv, glob, collections
per = collections.defaultdict(dict)
for f in glob.glob("$O/step/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        e = per[int(r["Dispatch_Id"])]
        e["name"] = r["Kernel_Name"]
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        e["us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
fam = collections.defaultdict(lambda: [0.0, 0.0, 0.0, 0])
def family(n):
    for key in ("conv3d_fwd_kernel", "conv3d_wgrad_kernel", "conv3d_fwd_rt_kernel", "conv1_fused_kernel", "conv1x1", "bn_", "xf_", "tok_", "slab_reduce", "heads", "adam"):
        if key in n:
            return key
    return "other"
for e in per.values():
    f = fam[family(e["name"])]
    f[0] += e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0); f[1] += e.get("GRBM_GUI_ACTIVE", 0.0) / 8.0; f[2] += e["us"]; f[3] += 1
tb = sum(f[0] for f in fam.values()); ta = sum(f[1] for f in fam.values()); tu = sum(f[2] for f in fam.values())
print("# tools/scripts/r04_pmc.sh (2): rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py --steps 10 --warmup 2 (+ 3 set-up steps, the roofline loop's launches)")
print("# every kernel of the process, serialised by the counter collection; busy = MFMA busy cycles / (1024 SIMDs x active cycles)")
for k, f in sorted(fam.items(), key=lambda kv: -kv[1][2]):
    print(f"{k:22s} launches {f[3]:6d}  time {f[2] / 1e3:9.2f} ms  clock {f[1] / max(f[2], 1e-9) / 1e3:5.2f} GHz  matrix pipe busy {f[0] / max(1024.0 * f[1], 1e-9):6.3f}")
print(f"{'ALL':22s} time {tu / 1e3:9.2f} ms  matrix pipe busy {tb / (1024.0 * ta):6.3f} of the active cycles (fp32 MFMA: 64 busy cycles per v_mfma_f32_32x32x2_f32)")
PY
cat $O/step_busy.txt; tail -2 $O/step.log
