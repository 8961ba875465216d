#!/usr/bin/env python3
"""Timing of the bf16 weight-gradient launches (bf16 tensors) of one encoder step at configs[2] sizes; run it once per
library (TMF_LIB=...) on the same box for an A/B.   python tools/wgrad_ab.py [--S 128]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transmf_ad_amd import ops                # noqa: E402
from tools.conv_ab import LAYERS              # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=8)
ap.add_argument("--S", type=int, default=128)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--fp32", action="store_true", help="the exact-fp32 weight gradient on fp32 tensors (peak 157.3 TF)")
a = ap.parse_args()
print("library:", os.environ.get("TMF_LIB", "in-tree"))
tot = 0.0
for name, cin, cout, k, div in LAYERS:
    if k != 3:
        continue
    s = a.S // div
    x = torch.randn((a.B, s, s, s, cin), device="cuda:0")
    dz = torch.randn((a.B, s, s, s, cout), device="cuda:0")
    if not a.fp32:
        x, dz = x.bfloat16(), dz.bfloat16()
    best = 1e9
    for _ in range(a.rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for i in range(a.reps + 2):
            if i == 2:
                e0.record()
            if a.fp32:
                ops.conv3d_wgrad(x, dz, cin, cout, 3)
            else:
                ops.conv3d_wgrad_bf16(x, dz, cin, cout)
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / a.reps)
    tot += best
    flop = 2.0 * a.B * s ** 3 * cin * cout * 27
    print(f"{name:8s} wgrad {best * 1e3:7.1f} us {flop / best / 1e9 / (157.3 if a.fp32 else 2500):5.3f}", flush=True)
print(f"sum {tot * 1e3:7.1f} us")
