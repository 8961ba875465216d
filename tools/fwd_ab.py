#!/usr/bin/env python3
"""A/B of the large-brick bf16 forward kernel, register-staged vs LDS-DMA form (tmf_set_option("bf16_dma", 0 | 1)), bf16
tensors, forward and data-gradient shapes of the sNet layers.   python tools/fwd_ab.py [--size 128] [--B 8]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transmf_ad_amd import _lib, ops          # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=128)
ap.add_argument("--B", type=int, default=8)
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
    s = a.size >> lvl
    fl = 2.0 * 27 * ci * co * a.B * s ** 3
    for tag, cin, cout in (("fwd", ci, co), ("dgrad", co, ci)):
        x = torch.randn((a.B, s, s, s, cin), device=dev).bfloat16()
        w = ops.pack_weight_bf16(torch.randn((cout, cin, 3, 3, 3), device=dev) * 0.05)
        row = []
        for dma in (0, 1):
            _lib.call("tmf_set_option", b"bf16_dma", dma)
            ms = t(lambda: ops.conv3d_bf16_raw(x, w, cin, cout, tag == "fwd", out_bf16=True))     # data gradients take no statistics
            row.append(f"{'dma' if dma else 'reg'} {ms * 1e3:7.1f} us {fl / ms / 1e9:6.0f} TF")
        k = _lib.query("tmf_conv3d_fwd_bf16_kernel_name", a.B, s, s, s, cin, cout, 3).decode()
        print(f"{name} {tag:5s} ({cin}->{cout} @{s}^3) {k}: " + " | ".join(row), flush=True)
_lib.call("tmf_set_option", b"bf16_dma", 1)
