#!/usr/bin/env python3
"""Timing ablations of the 8x8x8-brick bf16 forward kernel (tmf_set_option("debug", bits); results are garbage):
which part of a launch the waves spend waiting on.  The switches exist only in a -DTMF_ABLATE build:
    export TMF_EXTRA_FLAGS=-DTMF_ABLATE; python -m transmf_ad_amd.build; python tools/bf16_ablate.py [--ci 32 --co 64 --s 64]
(and rebuild without the variable afterwards)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transmf_ad_amd import _lib, ops          # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ci", type=int, default=32)
ap.add_argument("--co", type=int, default=64)
ap.add_argument("--s", type=int, default=64)
ap.add_argument("--B", type=int, default=8)
a = ap.parse_args()
if "-DTMF_ABLATE" not in os.environ.get("TMF_EXTRA_FLAGS", ""):
    sys.exit("bf16_ablate: needs a library built with TMF_EXTRA_FLAGS=-DTMF_ABLATE (see the header of this file)")
dev = "cuda:0"
x = torch.randn((a.B, a.s, a.s, a.s, a.ci), device=dev).bfloat16()
w = ops.pack_weight_bf16(torch.randn((a.co, a.ci, 3, 3, 3), device=dev) * 0.05)
_lib.call("tmf_set_option", b"bf16_v2", 2)
aa = torch.randn((4096, 4096), device=dev)
for _ in range(200):
    torch.mm(aa, aa)


def t(reps=20):
    for _ in range(3):
        ops.conv3d_bf16_raw(x, w, a.ci, a.co, True, out_bf16=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.conv3d_bf16_raw(x, w, a.ci, a.co, True, out_bf16=True)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


fl = 2.0 * 27 * a.ci * a.co * a.B * a.s ** 3
for bits, name in ((0, "full kernel"), (1, "no weight loads"), (2, "no halo loads"), (3, "no global loads"), (4, "no stage barriers"),
                   (8, "no MFMAs (and no LDS operand reads)"), (16, "no output stores"), (7, "no loads, no barriers"),
                   (23, "no loads, no barriers, no stores (MFMA + LDS stream only)"), (31, "nothing but the skeleton")):
    _lib.call("tmf_set_option", b"debug", bits)
    us = t()
    print(f"{name:58s} {us:8.1f} us   {fl / us / 1e6:7.0f} TF-equivalent", flush=True)
_lib.call("tmf_set_option", b"debug", 0)
