#!/usr/bin/env python3
"""A/B of the two bf16 weight-gradient kernels (tmf_set_option("wgrad_tr", 0 | 1)) on the sNet layer shapes.
   python tools/wgrad_ab.py [--size 128] [--B 8]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transmf_ad_amd import _lib, ops          # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=128)
ap.add_argument("--B", type=int, default=8)
ap.add_argument("--layer", default=None, help="one layer only, e.g. conv2.3 (for a kernel trace)")
a = ap.parse_args()
dev = "cuda:0"
aa = torch.randn((4096, 4096), device=dev)
for _ in range(100):
    torch.mm(aa, aa)


def t(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, ci, co, lvl in (("conv2.0", 32, 32, 1), ("conv2.3", 32, 64, 1), ("conv3.0", 64, 64, 2), ("conv3.3", 64, 128, 2),
                          ("conv4.0", 128, 256, 3)):
    if a.layer and a.layer != name:
        continue
    s = a.size >> lvl
    fl = 2.0 * 27 * ci * co * a.B * s ** 3
    x = torch.randn((a.B, s, s, s, ci), device=dev)
    dz = torch.randn((a.B, s, s, s, co), device=dev)
    x16, dz16 = x.bfloat16(), dz.bfloat16()
    row = []
    for tr in (0, 2):
        _lib.call("tmf_set_option", b"wgrad_tr", tr)
        for xx, dd, tag in ((x16, dz16, "bf16 tensors"), (x, dz, "fp32 tensors")):
            ms = t(lambda: ops.conv3d_wgrad_bf16(xx, dd, ci, co))
            row.append(f"{'tr' if tr else 'reg'} {tag} {ms * 1e3:7.1f} us {fl / ms / 1e9:6.0f} TF")
    print(f"{name} ({ci}->{co} @{s}^3): " + " | ".join(row), flush=True)
_lib.call("tmf_set_option", b"wgrad_tr", 1)
