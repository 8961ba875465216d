# LDS counters of the Winograd kernels (conflict-free slot map claim of DESIGN.md 3.15): one pass, --pmc only with --kernel-trace
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04pmcl
rm -rf $O; mkdir -p $O
cd $R
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_LDS -d $O/a -o a --output-format csv -- python3 tools/wino_check.py --only conv2 --no-time > $O/a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_LDS -d $O/b -o b --output-format csv -- python3 tools/wino_wgrad_check.py --only conv2 --no-time > $O/b.log 2>&1
python3 - <<PY > $O/lds.txt
import csv, glob, collections
print("# tools/scripts/r04_pmc_wino_lds.sh: rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_LDS over tools/wino_check.py / wino_wgrad_check.py --only conv2 --no-time (B = 2, 48^3)")
print("# per launch of the largest dispatches of each kernel: LDS-array cycles, of which bank-conflict cycles, LDS busy = active / (256 CUs x GPU cycles)")
per = collections.defaultdict(dict)
for f in glob.glob("$O/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = (f, int(r["Dispatch_Id"]))
        e = per[k]
        e["name"] = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        e["us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
best = {}
for e in per.values():
    if "wino" not in e["name"] and "conv3d_fwd_kernel" not in e["name"] and "conv3d_wgrad_kernel" not in e["name"]:
        continue
    if e["name"] not in best or e["us"] > best[e["name"]]["us"]:
        best[e["name"]] = e
for n, e in sorted(best.items()):
    act = e.get("SQ_LDS_IDX_ACTIVE", 0.0); conf = e.get("SQ_LDS_BANK_CONFLICT", 0.0); gui = e.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    print(f"{n[:70]:70s} {e['us']:8.1f} us  LDS active {act / 1e6:8.2f} M cycles  conflicts {conf / 1e6:7.2f} M ({100 * conf / max(act, 1):5.1f} %)  LDS busy {act / max(256 * gui, 1):5.3f}  LDS instructions {e.get('SQ_INSTS_LDS', 0) / 1e6:6.2f} M")
PY
cat $O/lds.txt
