#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter values per kernel over the dispatches of one run.
   python tools/pmc_summary.py OUT_DIR KERNEL_SUBSTRING [--skip 1]   -> JSON on stdout"""
import argparse
import csv
import glob
import json
import os
from collections import defaultdict

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("kernel")
ap.add_argument("--skip", type=int, default=1, help="leading dispatches to drop (cold caches)")
a = ap.parse_args()
vals = defaultdict(lambda: defaultdict(float))      # counter -> dispatch -> value
dur = {}
for f in glob.glob(os.path.join(a.dir, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if a.kernel not in r["Kernel_Name"]:
                continue
            d = int(r["Dispatch_Id"])
            vals[r["Counter_Name"]][d] += float(r["Counter_Value"])
            if "Start_Timestamp" in r and r["Start_Timestamp"]:
                dur[d] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
out = {}
for c, per in vals.items():
    ds = sorted(per)[a.skip:]
    if ds:
        out[c] = sum(per[d] for d in ds) / len(ds)
ds = sorted(dur)[a.skip:]
if ds:
    out["duration_us"] = sum(dur[d] for d in ds) / len(ds)
    out["dispatches"] = len(ds)
print(json.dumps(out, indent=1))
