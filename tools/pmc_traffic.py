#!/usr/bin/env python3
"""HBM traffic per launch of every conv kernel instance from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE,
separate runs) of `bench.py --roofline-only`:   python tools/pmc_traffic.py FETCH_DIR WRITE_DIR OUT.json [--note text]
bytes = FETCH_SIZE x 2 (gfx950: the counter reports half of a wide coalesced read, MI355X_MICROARCH.md) + WRITE_SIZE, both
in KiB in rocprofv3's output; averaged over all dispatches of an instance (= the launches the roofline object averages)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def collect(d, counter):
    per = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        disp = defaultdict(float)
        names = {}
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] != counter:
                    continue
                disp[int(r["Dispatch_Id"])] += float(r["Counter_Value"])
                names[int(r["Dispatch_Id"])] = r["Kernel_Name"]
        for k, v in disp.items():
            per[names[k]].append(v)
    return per


def short(name):
    n = re.sub(r"\(anonymous namespace\)::", "", name)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*$", "", n)


fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
note = sys.argv[5] if len(sys.argv) > 5 and sys.argv[4] == "--note" else ""
out = {}
for name, vals in fetch.items():
    s = short(name)
    if not any(k in s for k in ("conv3d_", "conv1_fused")):
        continue
    w = write.get(name, [])
    if not w:
        continue
    fb = sum(vals) / len(vals) * 1024.0 * 2.0
    wb = sum(w) / len(w) * 1024.0
    # the key bench.py matches: the instance text without the trailing VEC / FUSED template flags of the fp32 kernels
    key = re.sub(r">, (true|false), (true|false)>$", ">", s) if "FwdCfg" in s else (re.sub(r">, (true|false)>$", ">", s) if "WgCfg" in s else s)
    key = key.replace("conv3d_fwd_kernel<", "").replace("conv3d_wgrad_kernel<", "")
    out[key] = {"hbm_bytes_per_launch": round(fb + wb), "fetch_bytes_x2": round(fb), "write_bytes": round(wb),
                "dispatches": len(vals), "kernel": s,
                "note": "rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes of bench.py --roofline-only, average over "
                        "the instance's dispatches" + (": " + note if note else "")}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in out.items():
    print(f"{v['hbm_bytes_per_launch'] / 1e6:9.1f} MB  {v['dispatches']:3d}x  {k}")
