#!/usr/bin/env python3
"""Timing ablation of the wgrad kernel (debug bits: 1 = no staging after the first brick, 2 = no MFMA loop)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transmf_ad_amd import _lib, ops
from tools.kbench import timeit
dev = "cuda:0"
for (cin, cout, s) in ([] if os.environ.get("FWD_ONLY") else [(32, 64, 48), (32, 32, 48), (64, 64, 24), (128, 256, 12)]):
    x = torch.randn((8, s, s, s, cin), device=dev)
    dz = torch.randn((8, s, s, s, cout), device=dev)
    fl = 2.0 * 27 * cin * cout * 8 * s ** 3
    row = []
    for dbg in (0, 1, 2, 3):
        _lib.call("tmf_set_option", b"debug", dbg)
        ms = timeit(lambda: ops.conv3d_wgrad(x, dz, cin, cout, 3), 10)
        row.append(f"dbg{dbg}: {ms:.3f} ms ({fl / ms / 1e9:6.1f} TF)")
    _lib.call("tmf_set_option", b"debug", 0)
    print(f"wgrad {cin}->{cout} @{s}^3  " + " | ".join(row), flush=True)

# forward / dgrad kernel: bits 2 = no MFMA loop, 4 = no output stores
for (cin, cout, s_) in [(32, 64, 48), (32, 32, 48), (64, 32, 48), (64, 64, 24), (128, 256, 12)]:
    x = torch.randn((8, s_, s_, s_, cin), device=dev)
    w = torch.randn((27, cin, cout), device=dev) * 0.03
    fl = 2.0 * 27 * cin * cout * 8 * s_ ** 3
    row = []
    for dbg in (0, 2, 4, 6):            # 8 = no weight ring / stage barriers, 16 = no halo loads
        _lib.call("tmf_set_option", b"debug", dbg)
        ms = timeit(lambda: ops.conv3d_raw(x, w, cin, cout, 3, True), 30)
        row.append(f"dbg{dbg}: {ms:.3f} ms ({fl / ms / 1e9:6.1f} TF)")
    _lib.call("tmf_set_option", b"debug", 0)
    print(f"fwd {cin}->{cout} @{s_}^3  " + " | ".join(row), flush=True)
