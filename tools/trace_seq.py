#!/usr/bin/env python3
"""The launch sequence of the last train step of a rocprofv3 --kernel-trace CSV between two offsets into the step:

    python tools/trace_seq.py <kernel_trace.csv> [--from-ms 2.6] [--to-ms 4.4]

one line per kernel: start (ms into the step), duration (us), the idle time in front of it (us; over all streams), stream, name.
"""
import argparse
import csv

ap = argparse.ArgumentParser()
ap.add_argument("csv")
ap.add_argument("--from-ms", type=float, default=0.0)
ap.add_argument("--to-ms", type=float, default=1e9)
a = ap.parse_args()
rows = []
with open(a.csv) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", "")))
rows.sort()
adam = [r for r in rows if "adam_step_kernel" in r[2] or "FusedAdam" in r[2]]
ends = []
for r in adam:
    if ends and r[0] - ends[-1] < 1_000_000:
        ends[-1] = r[1]
    else:
        ends.append(r[1])
t0, t1 = ends[-2], ends[-1]
print(f"last step: {(t1 - t0) / 1e6:.3f} ms")
busy_until = t0
for s, e, name, stream in rows:
    if s < t0 or s > t1:
        continue
    gap = max(0, s - busy_until)
    off = (s - t0) / 1e6
    if a.from_ms <= off <= a.to_ms:
        short = name.replace("void ", "").replace("(anonymous namespace)::", "").replace("at::native::", "")[:70]
        print(f"{off:7.3f} ms  {(e - s) / 1e3:7.1f} us  gap {gap / 1e3:6.1f}  s{stream:>2}  {short}")
    busy_until = max(busy_until, e)
