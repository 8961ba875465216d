#!/usr/bin/env python3
"""Winograd conv kernel (csrc/conv3d_wino.hip) against an fp64 reference and the direct fp32 kernel, layer by layer:
max error relative to max |z|, statistic partials, and time of both kernels.
python tools/wino_check.py [--B 8 --S 96 --reps 10 --only conv2 --shapes 7,9,13]"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transmf_ad_amd import _lib, ops          # noqa: E402
from tools.conv_ab import LAYERS              # noqa: E402


def ref64(x, w):
    return F.conv3d(x.double().permute(0, 4, 1, 2, 3), w.double(), None, 1, 1).permute(0, 2, 3, 4, 1)


def check(B, D, H, W, cin, cout, dev, label):
    x = torch.randn((B, D, H, W, cin), device=dev)
    w = torch.randn((cout, cin, 3, 3, 3), device=dev) * (cin * 27) ** -0.5
    uf, ud = ops.pack_weights_wino(w, True, ops.wino_ok(cout, cin))
    z, part, nblk = ops.conv3d_wino_raw(x, uf, cin, cout, True)
    zd, partd, _ = ops.conv3d_raw(x, ops.pack_weight(w), cin, cout, 3, True)
    r = ref64(x, w)
    sc = r.abs().max().item()
    ew, ed = (z.double() - r).abs().max().item() / sc, (zd.double() - r).abs().max().item() / sc
    s1 = part[:, 0].double().sum(0)
    s2 = part[:, 1].double().sum(0)
    es1 = (s1 - z.double().sum((0, 1, 2, 3))).abs().max().item() / max(1.0, z.double().sum((0, 1, 2, 3)).abs().max().item())
    es2 = (s2 - (z.double() ** 2).sum((0, 1, 2, 3))).abs().max().item() / (z.double() ** 2).sum((0, 1, 2, 3)).abs().max().item()
    msg = f"{label:10s} {B}x{D}x{H}x{W} {cin:3d}->{cout:3d}  err wino {ew:.2e} direct {ed:.2e}  stats {es1:.1e} {es2:.1e}"
    if ud is not None:                                  # data gradient: dz [.., cout] -> dx [.., cin]
        dz = torch.randn((B, D, H, W, cout), device=dev)
        dx, _, _ = ops.conv3d_wino_raw(dz, ud, cout, cin, False)
        rd = F.conv_transpose3d(dz.double().permute(0, 4, 1, 2, 3), w.double(), None, 1, 1).permute(0, 2, 3, 4, 1)
        msg += f"  dgrad {(dx.double() - rd).abs().max().item() / rd.abs().max().item():.2e}"
    print(msg, flush=True)
    return ew


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--S", type=int, default=96)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--only", default="")
    ap.add_argument("--shapes", default="7,9,13")
    ap.add_argument("--no-time", action="store_true")
    a = ap.parse_args()
    dev = "cuda:0"
    torch.manual_seed(0)
    d, h, w = (int(v) for v in a.shapes.split(","))
    worst = check(2, d, h, w, 8, 32, dev, "ragged")
    worst = max(worst, check(1, 4, 8, 8, 16, 32, dev, "one brick"))
    worst = max(worst, check(2, 6, 10, 12, 32, 64, dev, "even"))
    worst = max(worst, check(8, 12, 12, 12, 32, 64, dev, "folded"))          # four samples x 4x4x4 bricks
    worst = max(worst, check(7, 11, 13, 11, 16, 32, dev, "folded rag"))
    worst = max(worst, check(3, 4, 4, 4, 8, 32, dev, "folded one"))
    tot = {"wino": 0.0, "direct": 0.0}
    for name, cin, cout, k, div in LAYERS:
        if k != 3 or (a.only and a.only not in name):
            continue
        s = a.S // div
        worst = max(worst, check(min(a.B, 2), s, s, s, cin, cout, dev, name))
        if a.no_time:
            continue
        for what in ("fwd", "dgrad"):
            ci, co = (cin, cout) if what == "fwd" else (cout, cin)
            if not ops.wino_ok(ci, co):
                continue
            x = torch.randn((a.B, s, s, s, ci), device=dev)
            wt = torch.randn((co, ci, 3, 3, 3), device=dev) * (ci * 27) ** -0.5
            uf, _ = ops.pack_weights_wino(wt, True, False)
            wp = ops.pack_weight(wt)
            best = {"wino": 1e9, "direct": 1e9}
            for _ in range(a.rounds):
                for v in ("wino", "direct"):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    for i in range(a.reps + 2):
                        if i == 2:
                            e0.record()
                        if v == "wino":
                            ops.conv3d_wino_raw(x, uf, ci, co, what == "fwd")
                        else:
                            ops.conv3d_raw(x, wp, ci, co, 3, what == "fwd")
                    e1.record()
                    e1.synchronize()
                    best[v] = min(best[v], e0.elapsed_time(e1) / a.reps)
            flop = 2.0 * a.B * s ** 3 * ci * co * 27
            for v in best:
                tot[v] += best[v]
            print(f"{name:8s} {what:5s} wino {best['wino'] * 1e3:7.1f} us ({flop / best['wino'] / 1e9 / 157.3:5.2f} of the direct-form peak)"
                  f"   direct {best['direct'] * 1e3:7.1f} us ({flop / best['direct'] / 1e9 / 157.3:5.3f})   x{best['direct'] / best['wino']:.2f}",
                  flush=True)
    print(f"sum wino {tot['wino'] * 1e3:.1f} us  direct {tot['direct'] * 1e3:.1f} us   worst error {worst:.2e}")
    return 0 if worst < 2e-5 else 1


if __name__ == "__main__":
    sys.exit(main())
