#!/usr/bin/env python3
"""A/B of the two bf16 forward kernels (4x8x8 bricks vs 8x8x8 bricks with 2 x NT register tiles) per layer shape.
   python tools/bf16_ab.py [--S 128] [--B 8] [--reps 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transmf_ad_amd import _lib, ops          # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--S", type=int, default=128)
ap.add_argument("--B", type=int, default=8)
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
dev = "cuda:0"


def t(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


aa = torch.randn((4096, 4096), device=dev)
for _ in range(200):
    torch.mm(aa, aa)
torch.cuda.synchronize()
for name, ci, co, lvl in (("conv2.0", 32, 32, 1), ("conv2.3", 32, 64, 1), ("conv2.3 dgrad", 64, 32, 1), ("conv3.0", 64, 64, 2),
                          ("conv3.3", 64, 128, 2), ("conv3.3 dgrad", 128, 64, 2), ("conv4.0", 128, 256, 3),
                          ("conv4.0 dgrad", 256, 128, 3)):
    s = a.S >> lvl
    fl = 2.0 * 27 * ci * co * a.B * s ** 3
    w = ops.pack_weight_bf16(torch.randn((co, ci, 3, 3, 3), device=dev) * (27 * ci) ** -0.5)
    row = [f"{name:14s} {s:3d}^3 {ci:3d}->{co:3d}"]
    for st16 in (True, False):
        x = torch.randn((a.B, s, s, s, ci), device=dev)
        if st16:
            x = x.bfloat16()
        outs = []
        for mode in (0, 2):
            _lib.call("tmf_set_option", b"bf16_v2", mode)
            ms = t(lambda: ops.conv3d_bf16_raw(x, w, ci, co, True, out_bf16=st16), a.reps)
            outs.append(ops.conv3d_bf16_raw(x, w, ci, co, True, out_bf16=st16)[0].float())
            row.append(f"{'bf16' if st16 else 'fp32'} tensors {'v2' if mode else 'v1'} {ms * 1e3:7.1f} us {fl / ms / 1e9:6.0f} TF")
        row.append(f"maxdiff {(outs[0] - outs[1]).abs().max().item():.1e}")
    print("  ".join(row), flush=True)
_lib.call("tmf_set_option", b"bf16_v2", 1)
