#!/usr/bin/env python3
"""A/B of the forward / data-gradient conv launches of one encoder step under tmf_set_option("conv_waves", v):
every (layer, direction) is timed with HIP events, the variants interleaved round by round so clock drift hits all
of them alike.   python tools/conv_ab.py [--B 8 --S 96 --waves 16,8,1 --rounds 6 --reps 10 --only conv2]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transmf_ad_amd import _lib, ops          # noqa: E402

# (name, cin, cout, kernel, volume divisor) of sNet(dim=128) behind the first block (networks.py:28-49)
LAYERS = [("conv2.0", 32, 32, 3, 2), ("conv2.3", 32, 64, 3, 2), ("conv3.0", 64, 64, 3, 4), ("conv3.3", 64, 128, 3, 4),
          ("conv4.0", 128, 256, 3, 8), ("conv4.3", 256, 128, 1, 8)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--S", type=int, default=96)
    ap.add_argument("--waves", default="16,8")
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--only", default="")
    ap.add_argument("--opt", default="conv_waves", help="the tmf_set_option name the variants are values of (conv_waves, conv_rt)")
    a = ap.parse_args()
    variants = [int(v) for v in a.waves.split(",")]
    dev = "cuda:0"
    for name, cin, cout, k, div in LAYERS:
        if a.only and a.only not in name:
            continue
        s = a.S // div
        x = torch.randn((a.B, s, s, s, cin), device=dev)
        dz = torch.randn((a.B, s, s, s, cout), device=dev)
        w = torch.randn((cout, cin, k, k, k), device=dev) * (cin * k ** 3) ** -0.5
        wp, wd = ops.pack_weight(w), ops.pack_weight_dgrad(w)
        flop = 2.0 * a.B * s ** 3 * cin * cout * k ** 3
        for what in ("fwd", "dgrad"):
            if what == "dgrad" and k == 1:
                continue
            best = {v: 1e9 for v in variants}
            for _ in range(a.rounds):
                for v in variants:
                    _lib.call("tmf_set_option", a.opt.encode(), v)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    for i in range(a.reps + 2):
                        if i == 2:
                            e0.record()
                        if what == "fwd":
                            ops.conv3d_raw(x, wp, cin, cout, k, True)
                        else:
                            ops.conv3d_raw(dz, wd, cout, cin, k, False)
                    e1.record()
                    e1.synchronize()
                    best[v] = min(best[v], e0.elapsed_time(e1) / a.reps)
            print(f"{name:8s} {what:5s} " + "  ".join(f"w{v}: {best[v] * 1e3:7.1f} us {flop / best[v] / 1e9 / 157.3:5.3f}" for v in variants),
                  flush=True)
    _lib.call("tmf_set_option", b"conv_waves", 16)
    _lib.call("tmf_set_option", b"conv_rt", 0)


if __name__ == "__main__":
    main()
