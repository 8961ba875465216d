#!/usr/bin/env python3
"""Launch the Winograd forward of one layer a few times (for rocprofv3 --pmc passes).
   python tools/winox_one.py LAYER [--B 8 --S 96 --reps 5 --x 1]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transmf_ad_amd import _lib, ops          # noqa: E402
from tools.conv_ab import LAYERS              # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("layer")
ap.add_argument("--B", type=int, default=8)
ap.add_argument("--S", type=int, default=96)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--x", type=int, default=1)
a = ap.parse_args()
name, cin, cout, k, div = [l for l in LAYERS if l[0] == a.layer][0]
s = a.S // div
_lib.call("tmf_set_option", b"wino_x", a.x)
x = torch.randn((a.B, s, s, s, cin), device="cuda:0")
w = torch.randn((cout, cin, 3, 3, 3), device="cuda:0") * (cin * 27) ** -0.5
uf, _ = ops.pack_weights_wino(w, True, False)
for _ in range(a.reps):
    ops.conv3d_wino_raw(x, uf, cin, cout, True)
torch.cuda.synchronize()
