#!/usr/bin/env python3
"""conv_persist A/B on the sNet layer shapes: one brick per workgroup vs persistent workgroups with a one-time stagger."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transmf_ad_amd import _lib, ops
from tools.kbench import timeit
dev = "cuda:0"
shapes = [(32, 32, 48), (32, 64, 48), (64, 64, 24), (64, 128, 24), (128, 256, 12)]
ws = torch.randn((27, 32, 32), device=dev) * 0.03
xs = torch.randn((8, 48, 48, 48, 32), device=dev)
for _ in range(600):
    ops.conv3d_raw(xs, ws, 32, 32, 3, True)
torch.cuda.synchronize()
for cin, cout, s in shapes:
    x = torch.randn((8, s, s, s, cin), device=dev)
    w = torch.randn((27, cin, cout), device=dev) * 0.03
    fl = 2.0 * 27 * cin * cout * 8 * s ** 3
    row = []
    ref = None
    for pv in (0, 1, 2, 3, 4, 6, 8, 0):
        _lib.call("tmf_set_option", b"conv_persist", pv)
        z, part, _ = ops.conv3d_raw(x, w, cin, cout, 3, True)
        if ref is None:
            ref = (z.clone(), part.sum(0).clone())
        else:
            assert torch.equal(z, ref[0]), "persistent result differs"
        ms = timeit(lambda: ops.conv3d_raw(x, w, cin, cout, 3, True), 40)
        row.append(f"p{pv}: {ms:.3f} ({fl / ms / 1e9:5.1f})")
    _lib.call("tmf_set_option", b"conv_persist", 0)
    print(f"fwd {cin}->{cout} @{s}^3  " + " | ".join(row), flush=True)
