#!/usr/bin/env python3
"""Kernel time per step by family from a rocprofv3 kernel_stats.csv of `bench.py --steps N` (TMF_STREAMS=1 for additive numbers).
python tools/kgroups.py <p_kernel_stats.csv> [more.csv ...]"""
import csv
import re
import sys

FAM = [("split fwd+dgrad (winox)", r"winox"), ("fp32 wgrad <0>", r"wino_wgrad_p_kernel<0>"), ("fp32 wgrad <1>", r"wino_wgrad_p_kernel<1>"),
       ("fp32 fwd/dgrad folded (wino_p)", r"wino_p_kernel"), ("wino other", r"wino_"), ("direct conv", r"conv3d_(fwd|wgrad)_kernel"),
       ("slab reduce", r"slab_reduce"), ("BatchNorm", r"bn_"), ("first block", r"conv1_fused|c1_"),
       ("token side", r"xf_|tok_|token_|layernorm|xattn"), ("1x1x1", r"conv1x1"), ("heads", r"heads"), ("adam", r"adam"),
       ("torch / runtime", r"at::|rocclr"), ("packs", r"pack")]
for path in sys.argv[1:]:
    rows = list(csv.DictReader(open(path)))
    steps = max([int(r["Calls"]) for r in rows if "adam_step_kernel" in r["Name"]] or [1])
    tot = {k: 0.0 for k, _ in FAM}
    tot["other"] = 0.0
    n = 0
    for r in rows:
        t = float(r["TotalDurationNs"]) / 1e6
        n += int(r["Calls"])
        for k, pat in FAM:
            if re.search(pat, r["Name"]):
                tot[k] += t
                break
        else:
            tot["other"] += t
    print(f"# {path}: {steps} steps, {n / steps:.1f} launches per step")
    for k, v in tot.items():
        if v > 0:
            print(f"{k:32s} {v / steps:7.3f} ms per step")
    print(f"{'ALL':32s} {sum(tot.values()) / steps:7.3f} ms per step")
