#!/usr/bin/env python3
"""GPU busy / idle accounting from a rocprofv3 --kernel-trace CSV (…_kernel_trace.csv).

    python tools/trace_gaps.py <kernel_trace.csv> [--last-ms 200] [--top 25]

Over the last `--last-ms` of the trace (the timed steps of bench.py): wall time, the UNION of the kernel intervals
(GPU busy with at least one kernel), idle time in gaps > 2 us, and the kernel-time table (sum over streams).  The
step is host-bound where the idle share is large; kernel-bound where busy ~ wall.
"""
import argparse
import csv
import collections
import re

ap = argparse.ArgumentParser()
ap.add_argument("csv")
ap.add_argument("--last-ms", type=float, default=0.0, help="window = the last N ms of the trace")
ap.add_argument("--steps", type=int, default=8, help="window = the last N train steps (delimited by the Adam launches)")
ap.add_argument("--top", type=int, default=25)
a = ap.parse_args()

rows = []
with open(a.csv) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", "")))
rows.sort()
if a.last_ms > 0:
    t_end = max(r[1] for r in rows)
    t0 = t_end - int(a.last_ms * 1e6)
else:
    # step boundaries: the end of each group of Adam launches (torch FusedAdam or adam_step_kernel) (groups are > 1 ms apart)
    adam = [r for r in rows if "FusedAdam" in r[2] or "adam_step_kernel" in r[2]]
    ends = []
    for r in adam:
        if ends and r[0] - ends[-1] < 1_000_000:
            ends[-1] = r[1]
        else:
            ends.append(r[1])
    t_end, t0 = ends[-1], ends[-1 - a.steps]
    print(f"{a.steps} steps: {(t_end - t0) / 1e6 / a.steps:.3f} ms per step")
rows = [r for r in rows if r[0] >= t0 and r[1] <= t_end]
wall = (t_end - t0) / 1e6
busy = 0
idle_gaps = []
cur_s, cur_e = rows[0][0], rows[0][1]
for s, e, _n, _st in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        idle_gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
# the largest idle gaps of ONE step (the last one in the window), with the kernels on either side
if a.last_ms <= 0:
    s0 = ends[-2]
    last = [r for r in rows if r[0] >= s0]
    gaps = []
    ce, prev = last[0][1], last[0][2]
    for s_, e_, n_, _st in last[1:]:
        if s_ > ce:
            gaps.append((s_ - ce, (ce - s0) / 1e6, prev, n_))
        if e_ >= ce:
            ce, prev = e_, n_
    short = lambda n: re.sub(r"\(.*", "", re.sub(r"\(anonymous namespace\)::|void |at::native::", "", n))[:48]
    print(f"last step: idle {sum(g[0] for g in gaps) / 1e6:.3f} ms in {len(gaps)} gaps; the 14 largest (us, at ms into the step, after -> before):")
    for g in sorted(gaps, reverse=True)[:14]:
        print(f"   {g[0] / 1e3:7.1f} us  @{g[1]:6.2f}  {short(g[2])}  ->  {short(g[3])}")
print(f"window {wall:.2f} ms: {len(rows)} launches, GPU busy (union) {busy / 1e6:.2f} ms = {busy / 1e6 / wall:.1%}, "
      f"idle {wall - busy / 1e6:.2f} ms in {len(idle_gaps)} gaps "
      f"({sum(1 for g in idle_gaps if g > 2000)} > 2 us: {sum(g for g in idle_gaps if g > 2000) / 1e6:.2f} ms, "
      f"{sum(1 for g in idle_gaps if g > 20000)} > 20 us: {sum(g for g in idle_gaps if g > 20000) / 1e6:.2f} ms)")
tab = collections.defaultdict(lambda: [0, 0])
for s, e, n, _st in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"\(.*", "", n)[:110]
    tab[n][0] += e - s
    tab[n][1] += 1
tot = sum(v[0] for v in tab.values())
print(f"kernel time (sum over streams) {tot / 1e6:.2f} ms")
for n, (t, c) in sorted(tab.items(), key=lambda kv: -kv[1][0])[:a.top]:
    print(f"  {t / 1e6:8.3f} ms {c:6d}x  {t / c / 1e3:8.1f} us  {n}")
