#!/usr/bin/env python3
"""The split Winograd kernel (csrc/conv3d_winox.hip: fp32 products as exact 3-way bf16 splits on the bf16 matrix pipe) against an
fp64 reference and against the fp32 Winograd kernel it replaces (tmf_set_option("wino_x", 0)): max error relative to max |z|,
statistic partials, the data gradient, run-to-run equality, the packed split weights bit for bit, and the time of both kernels.
python tools/winox_check.py [--B 8 --S 96 --reps 10 --only conv2 --no-time]"""
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


def setx(v):
    _lib.call("tmf_set_option", b"wino_x", v)


def split_of(uf, cin, cout):
    """The three bf16 parts the pack kernel wrote behind the fp32 tensor uf [64][cin/8][2][cout][4], as [3][64][cin][cout] floats."""
    n = 64 * cin * cout
    raw = torch.empty(0, dtype=torch.int16, device=uf.device).set_(uf.untyped_storage(), (uf.storage_offset() + n) * 2, (3 * n,))
    u3 = raw.view(64, cin // 16, 3, 2, cout, 8).to(torch.int32) << 16          # [p][c][t][half][co][j]
    f = u3.view(torch.float32)
    # K index: channel = 16 c + 8 s + 4 half + e, element j = 4 s + e
    f = f.view(64, cin // 16, 3, 2, cout, 2, 4).permute(2, 0, 1, 5, 3, 6, 4)  # [t][p][c][s][half][e][co]
    return f.reshape(3, 64, cin, cout)


def check(B, D, H, W, cin, cout, dev, label, check_pack=False):
    x = torch.randn((B, D, H, W, cin), device=dev)
    w = torch.randn((cout, cin, 3, 3, 3), device=dev) * (cin * 27) ** -0.5
    uf, ud = ops.pack_weights_wino(w, True, True)
    name = _lib.query("tmf_conv3d_wino_kernel_name2", B, D, H, W, cin, cout, 1).decode()
    setx(1)
    z, part, nblk = ops.conv3d_wino_raw(x, uf, cin, cout, True)
    z2, part2, _ = ops.conv3d_wino_raw(x, uf, cin, cout, True)
    z0_, _, _ = ops.conv3d_wino_raw(x, uf, cin, cout, False)
    setx(0)
    zf, partf, _ = ops.conv3d_wino_raw(x, uf, cin, cout, True)
    setx(1)
    r = ref64(x, w)
    sc = r.abs().max().item()
    ex, ef = (z.double() - r).abs().max().item() / sc, (zf.double() - r).abs().max().item() / sc
    exf = (z - zf).abs().max().item() / sc
    same = bool((z == z2).all()) and bool((part == part2).all()) and bool((z == z0_).all())
    zs = z.double()
    s1, s2 = part[:, 0].double().sum(0), part[:, 1].double().sum(0)
    es1 = (s1 - zs.sum((0, 1, 2, 3))).abs().max().item() / max(1.0, zs.sum((0, 1, 2, 3)).abs().max().item())
    es2 = (s2 - (zs ** 2).sum((0, 1, 2, 3))).abs().max().item() / (zs ** 2).sum((0, 1, 2, 3)).abs().max().item()
    msg = (f"{label:10s} {B}x{D}x{H}x{W} {cin:3d}->{cout:3d} [{name[7:22]}] err split {ex:.2e} fp32 {ef:.2e} split-fp32 {exf:.2e}"
           f"  stats {es1:.1e} {es2:.1e}  rerun {'same' if same else 'DIFFERS'}")
    dz = torch.randn((B, D, H, W, cout), device=dev)
    dx, _, _ = ops.conv3d_wino_raw(dz, ud, cout, cin, False)
    rd = F.conv_transpose3d(dz.double().permute(0, 4, 1, 2, 3), w.double(), None, 1, 1).permute(0, 2, 3, 4, 1)
    ed = (dx.double() - rd).abs().max().item() / rd.abs().max().item()
    msg += f"  dgrad {ed:.2e}"
    if check_pack:
        for u, ci, co in ((uf, cin, cout), (ud, cout, cin)):
            if ci % 32:
                continue
            parts = split_of(u, ci, co)
            u32 = u.permute(0, 1, 2, 4, 3).reshape(64, ci, co)              # [p][g][hs][co][e] -> [p][g][hs][e][co] = channel order
            exact = bool((parts.double().sum(0) == u32.double()).all())
            bf = bool(((parts.view(torch.int32) & 0xFFFF) == 0).all())
            msg += f"  split {'exact' if exact and bf else 'WRONG'}"
            if not (exact and bf):
                ex = 1.0
    print(msg, flush=True)
    return max(ex, ed) if same else 1.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--S", type=int, default=96)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--only", default="")
    ap.add_argument("--no-time", action="store_true")
    ap.add_argument("--time-only", action="store_true", help="skip every comparison (ablation builds give wrong results on purpose)")
    a = ap.parse_args()
    dev = "cuda:0"
    torch.manual_seed(0)
    worst = 0.0
    if not a.time_only:
        worst = check(1, 4, 8, 8, 32, 32, dev, "one brick", check_pack=True)
        worst = max(worst, check(2, 7, 9, 13, 32, 32, dev, "ragged"))
        worst = max(worst, check(1, 8, 16, 16, 32, 64, dev, "8 bricks", check_pack=True))
        worst = max(worst, check(3, 6, 10, 12, 64, 32, dev, "3 samples"))
        worst = max(worst, check(2, 12, 24, 24, 64, 128, dev, "groups"))
        worst = max(worst, check(2, 9, 17, 21, 128, 64, dev, "rag 8 ch"))
    tot = {"split": 0.0, "fp32": 0.0}
    for name, cin, cout, k, div in LAYERS:
        if k != 3 or (a.only and a.only not in name):
            continue
        s = a.S // div
        if not a.time_only:
            worst = max(worst, check(min(a.B, 2), s, s, s, cin, cout, dev, name))
        if a.no_time:
            continue
        for what in ("fwd", "dgrad"):
            ci, co = (cin, cout) if what == "fwd" else (cout, cin)
            x = torch.randn((a.B, s, s, s, ci), device=dev)
            wt = torch.randn((co, ci, 3, 3, 3), device=dev) * (ci * 27) ** -0.5
            uf, _ = ops.pack_weights_wino(wt, True, False)
            best = {"split": 1e9, "fp32": 1e9}
            for _ in range(a.rounds):
                for v in ("split", "fp32"):
                    setx(1 if v == "split" else 0)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    for i in range(a.reps + 2):
                        if i == 2:
                            e0.record()
                        ops.conv3d_wino_raw(x, uf, ci, co, what == "fwd")
                    e1.record()
                    e1.synchronize()
                    best[v] = min(best[v], e0.elapsed_time(e1) / a.reps)
            setx(1)
            kn = _lib.query("tmf_conv3d_wino_kernel_name2", a.B, s, s, s, ci, co, 1 if what == "fwd" else 0).decode()
            for v in best:
                tot[v] += best[v]
            print(f"{name:8s} {what:5s} [{kn[7:22]}] split {best['split'] * 1e3:7.1f} us   fp32 {best['fp32'] * 1e3:7.1f} us   x{best['fp32'] / best['split']:.2f}",
                  flush=True)
    print(f"sum split {tot['split'] * 1e3:.1f} us  fp32 {tot['fp32'] * 1e3:.1f} us   worst error {worst:.2e}")
    return 0 if worst < 2e-5 else 1


if __name__ == "__main__":
    sys.exit(main())
