#!/usr/bin/env python3
"""Winograd weight-gradient kernel against fp64 torch (small shapes) and the direct fp32 kernel (layer sizes), with timings.
python tools/wino_wgrad_check.py [--B 8 --S 96 --reps 10 --only conv2]"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transmf_ad_amd import _lib, ops          # noqa: E402
from tools.conv_ab import LAYERS              # noqa: E402


def check(B, D, H, W, cin, cout, dev, label, ref64=True):
    x = torch.randn((B, D, H, W, cin), device=dev)
    dz = torch.randn((B, D, H, W, cout), device=dev)
    dw = ops.conv3d_wgrad_wino(x, dz, cin, cout, reference_layout=True)
    dd = ops.conv3d_wgrad(x, dz, cin, cout, 3, reference_layout=True)
    msg = f"{label:10s} {B}x{D}x{H}x{W} {cin:3d}->{cout:3d}  wino vs direct {(dw - dd).abs().max().item() / dd.abs().max().item():.2e}"
    err = (dw - dd).abs().max().item() / dd.abs().max().item()
    if ref64:
        r = torch.nn.grad.conv3d_weight(x.double().permute(0, 4, 1, 2, 3), (cout, cin, 3, 3, 3), dz.double().permute(0, 4, 1, 2, 3),
                                        stride=1, padding=1)
        sc = r.abs().max().item()
        err = (dw.double() - r).abs().max().item() / sc
        msg += f"  vs fp64: wino {err:.2e} direct {(dd.double() - r).abs().max().item() / sc:.2e}"
    dt = ops.conv3d_wgrad_wino(x, dz, cin, cout, reference_layout=False)
    assert torch.equal(dt.view(3, 3, 3, cin, cout).permute(4, 3, 0, 1, 2).contiguous(), dw), "tap-major layout"
    assert torch.equal(dw, ops.conv3d_wgrad_wino(x, dz, cin, cout, reference_layout=True)), "run-to-run"
    print(msg, flush=True)
    return err


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--S", type=int, default=96)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--only", default="")
    ap.add_argument("--no-time", action="store_true")
    a = ap.parse_args()
    dev = "cuda:0"
    torch.manual_seed(0)
    worst = check(1, 4, 4, 8, 32, 32, dev, "one stage")
    worst = max(worst, check(2, 7, 9, 13, 32, 64, dev, "ragged"))
    worst = max(worst, check(2, 6, 10, 12, 64, 32, dev, "even"))
    tot = {"wino": 0.0, "direct": 0.0}
    for name, cin, cout, k, div in LAYERS:
        if k != 3 or (a.only and a.only not in name):
            continue
        s = a.S // div
        worst = max(worst, check(2, s, s, s, cin, cout, dev, name, ref64=s <= 24))
        if a.no_time:
            continue
        x = torch.randn((a.B, s, s, s, cin), device=dev)
        dz = torch.randn((a.B, s, s, s, cout), device=dev)
        best = {"wino": 1e9, "direct": 1e9}
        for _ in range(a.rounds):
            for v in ("wino", "direct"):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for i in range(a.reps + 2):
                    if i == 2:
                        e0.record()
                    if v == "wino":
                        ops.conv3d_wgrad_wino(x, dz, cin, cout, reference_layout=True)
                    else:
                        ops.conv3d_wgrad(x, dz, cin, cout, 3, reference_layout=True)
                e1.record()
                e1.synchronize()
                best[v] = min(best[v], e0.elapsed_time(e1) / a.reps)
        for v in best:
            tot[v] += best[v]
        flop = 2.0 * a.B * s ** 3 * cin * cout * 27
        print(f"{name:8s} wgrad wino {best['wino'] * 1e3:7.1f} us ({flop / best['wino'] / 1e9 / 157.3:5.2f})   direct {best['direct'] * 1e3:7.1f} us "
              f"({flop / best['direct'] / 1e9 / 157.3:5.3f})   x{best['direct'] / best['wino']:.2f}", flush=True)
    print(f"sum wino {tot['wino'] * 1e3:.1f} us  direct {tot['direct'] * 1e3:.1f} us   worst error {worst:.2e}")


if __name__ == "__main__":
    main()
