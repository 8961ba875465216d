#!/usr/bin/env python3
"""How much of the bf16 forward kernels' distance from the matrix peak is CLOCK (the chip clocks to its power budget,
MI355X_MICROARCH.md "DVFS give-back"): the same launches on random, sign-constant and zero operands.
    python tools/bf16_data_probe.py [--S 64]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transmf_ad_amd import _lib, ops          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--S", type=int, default=64)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=5)
    a = ap.parse_args()
    dev = "cuda:0"
    for cin, cout in ((32, 32), (32, 64), (64, 32)):
        fl = 2.0 * a.B * a.S ** 3 * cin * cout * 27
        fills = {"randn": lambda *s: torch.randn(s, device=dev), "abs": lambda *s: torch.randn(s, device=dev).abs(),
                 "zeros": lambda *s: torch.zeros(s, device=dev)}
        for _once in (0,):
            name = _lib.query("tmf_conv3d_fwd_bf16_kernel_name", a.B, a.S, a.S, a.S, cin, cout, 3).decode()
            row = []
            for fname, fill in fills.items():
                x = fill(a.B, a.S, a.S, a.S, cin).bfloat16()
                w = ops.pack_weight_bf16(fill(cout, cin, 3, 3, 3) * (27 * cin) ** -0.5)
                best = 1e9
                for _ in range(a.rounds):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    for i in range(a.reps + 3):
                        if i == 3:
                            e0.record()
                        ops.conv3d_bf16_raw(x, w, cin, cout, True, out_bf16=True)
                    e1.record()
                    e1.synchronize()
                    best = min(best, e0.elapsed_time(e1) / a.reps)
                row.append(f"{fname} {best * 1e3:6.1f} us {fl / best / 1e9 / 2500:5.3f}")
            print(f"{cin:3d}->{cout:3d} {name[:44]:44s} " + "   ".join(row), flush=True)


def split_probe(a):
    """the fp32x mode's split kernel (six bf16 products per fp32 product) on random and zero operands, 96^3 layer sizes"""
    dev = "cuda:0"
    for cin, cout, S in ((32, 32, 48), (32, 64, 48), (64, 128, 24)):
        fl = 2.0 * a.B * S ** 3 * cin * cout * 27
        row = []
        for fname, gen in (("randn", torch.randn), ("zeros", torch.zeros)):
            x = gen((a.B, S, S, S, cin), device=dev)
            w3, _ = ops.pack_weights_split3(gen((cout, cin, 3, 3, 3), device=dev) * (27 * cin) ** -0.5, False)
            best = 1e9
            for _ in range(a.rounds):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for i in range(a.reps + 3):
                    if i == 3:
                        e0.record()
                    ops.conv3d_split_raw(x, w3, cin, cout, True)
                e1.record()
                e1.synchronize()
                best = min(best, e0.elapsed_time(e1) / a.reps)
            row.append(f"{fname} {best * 1e3:6.1f} us {fl / best / 1e9:6.1f} TF (fp32 flops) = {6 * fl / best / 1e9 / 2500:5.3f} of the bf16 peak")
        print(f"split {cin:3d}->{cout:3d} @{S}^3  " + "   ".join(row), flush=True)


if __name__ == "__main__":
    if "--split" in sys.argv:
        sys.argv.remove("--split")
        ap = argparse.ArgumentParser()
        ap.add_argument("--B", type=int, default=8); ap.add_argument("--reps", type=int, default=20); ap.add_argument("--rounds", type=int, default=5)
        split_probe(ap.parse_args())
        sys.exit(0)
    main()
