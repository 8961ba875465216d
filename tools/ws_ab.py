#!/usr/bin/env python3
"""Wave-specialised forward kernel (tmf_set_option("conv_ws", 1)) against the product kernel: results and time."""
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
for B, cin, cout, shape in [(8, 32, 64, (48, 48, 48)), (8, 64, 64, (24, 24, 24)), (8, 64, 128, (24, 24, 24)), (8, 128, 64, (24, 24, 24)),
                            (2, 32, 64, (19, 21, 17)), (8, 32, 64, (64, 64, 64))]:
    D, H, W = shape
    x = torch.randn((B, D, H, W, cin), device=dev)
    w = torch.randn((27, cin, cout), device=dev) * (27 * cin) ** -0.5
    fl = 2.0 * 27 * cin * cout * B * D * H * W
    z0, p0, _ = ops.conv3d_raw(x, w, cin, cout, 3, True)
    t0 = timeit(lambda: ops.conv3d_raw(x, w, cin, cout, 3, True), 30)
    _lib.call("tmf_set_option", b"conv_ws", 1)
    z1, p1, _ = ops.conv3d_raw(x, w, cin, cout, 3, True)
    t1 = timeit(lambda: ops.conv3d_raw(x, w, cin, cout, 3, True), 30)
    _lib.call("tmf_set_option", b"conv_ws", 0)
    ref = torch.nn.functional.conv3d(x.permute(0, 4, 1, 2, 3).double(), w.view(3, 3, 3, cin, cout).permute(4, 3, 0, 1, 2).double(),
                                     padding=1).permute(0, 2, 3, 4, 1) if D <= 24 else None
    e01 = ((z1 - z0).abs().max() / z0.abs().max()).item()
    es = ((p1.double().sum(0) - p0.double().sum(0)).abs().max() / p0.double().sum(0).abs().max()).item()
    er = "" if ref is None else f" vs fp64: ring {((z0.double() - ref).abs().max() / ref.abs().max()).item():.1e} ws {((z1.double() - ref).abs().max() / ref.abs().max()).item():.1e}"
    print(f"fwd {cin}->{cout} @{shape} B={B}  ring {t0:.3f} ms ({fl / t0 / 1e9:5.1f} TF) | ws {t1:.3f} ms ({fl / t1 / 1e9:5.1f} TF)  "
          f"diff {e01:.1e} stats {es:.1e}{er}", flush=True)
