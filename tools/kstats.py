#!/usr/bin/env python3
"""Rows of a rocprofv3 kernel_stats.csv whose kernel name contains one of the given substrings: calls, average and total time.
python tools/kstats.py <p_kernel_stats.csv> [needle ...]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
needles = sys.argv[2:]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows:
    if not needles or any(n in r["Name"] for n in needles):
        print(f"{r['Name'][:86]:86s} {int(r['Calls']):5d} x {float(r['AverageNs']) / 1e3:8.1f} us = {float(r['TotalDurationNs']) / 1e6:8.2f} ms "
              f"({float(r['TotalDurationNs']) / tot * 100:5.2f} %)")
print(f"all kernels: {tot / 1e6:.2f} ms")
