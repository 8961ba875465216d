#!/usr/bin/env python3
"""Timing ablations of the wave-specialised forward kernel: conv_ws = 1 + 2 * bits (1: producers skip global loads,
2: skip LDS commits, 4: compute waves skip the z stores).  Results are wrong by construction; only the time is read."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transmf_ad_amd import _lib, ops
from tools.kbench import timeit
dev = "cuda:0"
ws = torch.randn((27, 32, 32), device=dev) * 0.03
xs = torch.randn((8, 48, 48, 48, 32), device=dev)
for _ in range(600):
    ops.conv3d_raw(xs, ws, 32, 32, 3, True)
torch.cuda.synchronize()
for B, cin, cout, shape in [(8, 32, 64, (48, 48, 48)), (8, 64, 64, (48, 48, 48)), (8, 64, 128, (24, 24, 24))]:
    D, H, W = shape
    x = torch.randn((B, D, H, W, cin), device=dev)
    w = torch.randn((27, cin, cout), device=dev) * (27 * cin) ** -0.5
    fl = 2.0 * 27 * cin * cout * B * D * H * W
    t0 = timeit(lambda: ops.conv3d_raw(x, w, cin, cout, 3, True), 30)
    out = f"{cin}->{cout}@{shape}: ring {fl / t0 / 1e9:5.1f}"
    for dbg, nm in [(0, "ws"), (1, "no global loads"), (2, "no LDS commits"), (3, "idle producers"), (4, "no stores"), (7, "MFMA + barriers only")]:
        _lib.call("tmf_set_option", b"conv_ws", 1 + 2 * dbg)
        t1 = timeit(lambda: ops.conv3d_raw(x, w, cin, cout, 3, True), 30)
        out += f" | {nm} {fl / t1 / 1e9:5.1f}"
    _lib.call("tmf_set_option", b"conv_ws", 0)
    print(out, flush=True)
