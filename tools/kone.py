#!/usr/bin/env python3
"""Launch ONE conv kernel variant a few times (for rocprofv3 --pmc passes).
   python tools/kone.py fwd|dgrad|wgrad LAYER [--B 8 --S 96 --reps 5 --waves 8]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transmf_ad_amd import _lib, ops          # noqa: E402
from tools.conv_ab import LAYERS              # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("what")
ap.add_argument("layer")
ap.add_argument("--B", type=int, default=8)
ap.add_argument("--S", type=int, default=96)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--waves", type=int, default=16)
ap.add_argument("--spin", type=int, default=0, help="launches of another conv shape first (loaded clock state)")
ap.add_argument("--fill", default="randn", help="operand data: randn | zeros (the chip clocks to its power budget: DVFS)")
ap.add_argument("--v2", type=int, default=1, help="tmf_set_option('bf16_v2', v): 0 small-brick, 2 8x8x8-brick bf16 forward kernel")
a = ap.parse_args()
_lib.call("tmf_set_option", b"conv_waves", a.waves)
_lib.call("tmf_set_option", b"bf16_v2", a.v2)
name, cin, cout, k, div = [l for l in LAYERS if l[0] == a.layer][0]
s = a.S // div
dev = "cuda:0"
gen = torch.zeros if a.fill == "zeros" else torch.randn
x = gen((a.B, s, s, s, cin), device=dev)
w = gen((cout, cin, k, k, k), device=dev) * (cin * k ** 3) ** -0.5
dz = gen((a.B, s, s, s, cout), device=dev)
wp, wd = ops.pack_weight(w), ops.pack_weight_dgrad(w)
if a.spin:
    ws = torch.randn((27, 32, 32), device=dev) * 0.03
    xs = torch.randn((a.B, 48, 48, 48, 32), device=dev)
    for _ in range(a.spin):
        ops.conv3d_raw(xs, ws, 32, 32, 3, True)
    torch.cuda.synchronize()
x16, dz16, wb16 = x.bfloat16(), dz.bfloat16(), ops.pack_weight_bf16(w)
for _ in range(a.reps):
    if a.what == "split":
        ops.conv3d_split_raw(x, ops.split3_bf16(w.permute(2, 3, 4, 0, 1).contiguous()), cin, cout, True)
    elif a.what == "bf16":
        ops.conv3d_bf16_raw(x, ops.pack_weight_bf16(w), cin, cout, True)
    elif a.what == "bf16s":
        ops.conv3d_bf16_raw(x16, wb16, cin, cout, True, out_bf16=True)
    elif a.what == "wgrad16s":
        ops.conv3d_wgrad_bf16(x16, dz16, cin, cout)
    elif a.what == "wgrad16":
        ops.conv3d_wgrad_bf16(x, dz, cin, cout)
    elif a.what == "fwd":
        ops.conv3d_raw(x, wp, cin, cout, k, True)
    elif a.what == "dgrad":
        ops.conv3d_raw(dz, wd, cout, cin, k, False)
    else:
        ops.conv3d_wgrad(x, dz, cin, cout, k)
torch.cuda.synchronize()
print("done", a.what, a.layer)
