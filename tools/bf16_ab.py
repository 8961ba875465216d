#!/usr/bin/env python3
"""A/B of the bf16 forward / data-gradient launches (bf16 tensors) of one encoder step at configs[2] sizes under
tmf_set_option("debug", bits): variants interleaved, best of rounds.  python tools/bf16_ab.py [--S 128 --dbg 0,32]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transmf_ad_amd import _lib, ops          # noqa: E402
from tools.conv_ab import LAYERS              # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--S", type=int, default=128)
    ap.add_argument("--dbg", default="0,32")
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--only", default="")
    ap.add_argument("--opt", default="debug", help="the tmf_set_option name the variants are values of (debug, bf16_w4)")
    a = ap.parse_args()
    print("library:", os.environ.get("TMF_LIB", "in-tree"))
    variants = [int(v) for v in a.dbg.split(",")]
    dev = "cuda:0"
    tot = {v: 0.0 for v in variants}
    for name, cin, cout, k, div in LAYERS:
        if k != 3 or (a.only and a.only not in name):
            continue
        s = a.S // div
        x = torch.randn((a.B, s, s, s, cin), device=dev).bfloat16()
        dz = torch.randn((a.B, s, s, s, cout), device=dev).bfloat16()
        w = torch.randn((cout, cin, k, k, k), device=dev) * (cin * k ** 3) ** -0.5
        wp, wd = ops.pack_weight_bf16(w), ops.pack_weight_dgrad_bf16(w)
        flop = 2.0 * a.B * s ** 3 * cin * cout * k ** 3
        for what in ("fwd", "dgrad"):
            best = {v: 1e9 for v in variants}
            for _ in range(a.rounds):
                for v in variants:
                    _lib.call("tmf_set_option", a.opt.encode(), v)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    for i in range(a.reps + 2):
                        if i == 2:
                            e0.record()
                        if what == "fwd":
                            ops.conv3d_bf16_raw(x, wp, cin, cout, True, out_bf16=True)
                        else:
                            ops.conv3d_bf16_raw(dz, wd, cout, cin, False, out_bf16=True)
                    e1.record()
                    e1.synchronize()
                    best[v] = min(best[v], e0.elapsed_time(e1) / a.reps)
            for v in variants:
                tot[v] += best[v]
            print(f"{name:8s} {what:5s} " + "  ".join(f"d{v}: {best[v] * 1e3:7.1f} us {flop / best[v] / 1e9 / 2500:5.3f}" for v in variants),
                  flush=True)
    print("sum      " + "  ".join(f"d{v}: {tot[v] * 1e3:7.1f} us" for v in variants))
    _lib.call("tmf_set_option", a.opt.encode(), 0)


if __name__ == "__main__":
    main()
