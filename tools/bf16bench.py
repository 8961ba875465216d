#!/usr/bin/env python3
"""Time the bf16 matrix-core conv against the fp32 one on the sNet layer shapes (forward launch, B=8, 96^3 input)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transmf_ad_amd import ops
from tools.kbench import LAYERS, timeit
dev = "cuda:0"
for name, cin, cout, k, div, pool in LAYERS:
    if cin == 1 or k != 3:
        continue
    s = 96 // div
    x = torch.randn((8, s, s, s, cin), device=dev)
    w = torch.randn((cout, cin, 3, 3, 3), device=dev) * (27 * cin) ** -0.5
    wp, wb = ops.pack_weight(w), ops.pack_weight_bf16(w)
    fl = 2.0 * 27 * cin * cout * 8 * s ** 3
    t32 = timeit(lambda: ops.conv3d_raw(x, wp, cin, cout, 3, True), 10)
    t16 = timeit(lambda: ops.conv3d_bf16_raw(x, wb, cin, cout, True), 10)
    w3 = ops.split3_bf16(w.permute(2, 3, 4, 0, 1).contiguous())
    t3 = timeit(lambda: ops.conv3d_split_raw(x, w3, cin, cout, True), 10)
    gb = (x.numel() + 8 * s ** 3 * cout) * 4 / 1e9
    print(f"{name}: fp32 {t32:.3f} ms ({fl/t32/1e9:6.1f} TF)   bf16 {t16:.3f} ms ({fl/t16/1e9:6.1f} TF, {gb/t16*1e3:5.0f} GB/s algorithmic)   fp32x(split) {t3:.3f} ms ({fl/t3/1e9:6.1f} TF-equivalent)", flush=True)
    dz = torch.randn((8, s, s, s, cout), device=dev)
    tw32 = timeit(lambda: ops.conv3d_wgrad(x, dz, cin, cout, 3), 10)
    tw16 = timeit(lambda: ops.conv3d_wgrad_bf16(x, dz, cin, cout), 10)
    print(f"{name}: wgrad fp32 {tw32:.3f} ms ({fl/tw32/1e9:6.1f} TF)   bf16 {tw16:.3f} ms ({fl/tw16/1e9:6.1f} TF)", flush=True)
