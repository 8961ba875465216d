#!/usr/bin/env python3
"""CPU experiment: how far does a Winograd F(2x2x2, 3x3x3) convolution in fp32 move the model's outputs?

The oracle's forward is run three times on a golden fixture — fp64 (truth), fp32 direct (torch's conv3d) and fp32 with
every 3x3x3 convolution behind the first block replaced by the Winograd form (transforms and the 64 per-position
products in fp32, the channel sum as an fp32 matmul) — and the logits / loss / feature probes are compared with the
fixture's fp64 values.  The GPU kernel has to stay inside the 1e-3 gate of BASELINE.json's north_star on every fixture,
the ill-conditioned batch-2 ones included.      python tools/winograd_numerics.py ad_full_b2 [--layers fwd]
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _golden import Golden, run_oracle      # noqa: E402

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def winograd_conv3d(x, w):
    """x [B, C, D, H, W], w [O, C, 3, 3, 3], padding 1, even D/H/W are padded up.  All arithmetic in x.dtype."""
    dt = x.dtype
    bt, g, at = BT.to(dt), G.to(dt), AT.to(dt)
    B, C, D, H, W = x.shape
    O = w.shape[0]
    D2, H2, W2 = (D + 1) // 2, (H + 1) // 2, (W + 1) // 2
    xp = F.pad(x, (1, 1 + 2 * W2 - W, 1, 1 + 2 * H2 - H, 1, 1 + 2 * D2 - D))
    # tiles: [B, C, D2, H2, W2, 4, 4, 4]
    t = xp.unfold(2, 4, 2).unfold(3, 4, 2).unfold(4, 4, 2)
    # input transform along the three tile axes, one axis at a time (the order of the kernel: d, h, w)
    v = torch.einsum("pa,bcxyzaij->bcxyzpij", bt, t)
    v = torch.einsum("qi,bcxyzpij->bcxyzpqj", bt, v)
    v = torch.einsum("rj,bcxyzpqj->bcxyzpqr", bt, v)
    u = torch.einsum("pa,ocaij->ocpij", g, w)
    u = torch.einsum("qi,ocpij->ocpqj", g, u)
    u = torch.einsum("rj,ocpqj->ocpqr", g, u)
    # 64 products: m[b, o, tile, p, q, r] = sum_c v * u   (an fp32 matmul per position)
    v2 = v.permute(5, 6, 7, 0, 2, 3, 4, 1).reshape(64, -1, C)
    u2 = u.permute(2, 3, 4, 1, 0).reshape(64, C, O)
    m = torch.bmm(v2, u2).reshape(4, 4, 4, B, D2, H2, W2, O)
    # output transform: w, then h, then d
    y = torch.einsum("kr,pqrbxyzo->pqkbxyzo", at, m)
    y = torch.einsum("jq,pqkbxyzo->pjkbxyzo", at, y)
    y = torch.einsum("ip,pjkbxyzo->ijkbxyzo", at, y)
    y = y.permute(3, 7, 4, 0, 5, 1, 6, 2).reshape(B, O, 2 * D2, 2 * H2, 2 * W2)
    return y[:, :, :D, :H, :W].contiguous()


class _WinoFn(torch.autograd.Function):
    """forward and / or data gradient in the Winograd form; the weight gradient stays direct."""
    @staticmethod
    def forward(ctx, x, w, mode):
        ctx.save_for_backward(x, w)
        ctx.mode = mode
        return winograd_conv3d(x, w) if "fwd" in mode else _orig(x, w, None, 1, 1)

    @staticmethod
    def backward(ctx, dz):
        x, w = ctx.saved_tensors
        if "dgrad" in ctx.mode:
            dx = winograd_conv3d(dz, w.flip(2, 3, 4).transpose(0, 1).contiguous())
        else:
            dx = torch.nn.grad.conv3d_input(x.shape, w, dz, stride=1, padding=1)
        dw = torch.nn.grad.conv3d_weight(x, w.shape, dz, stride=1, padding=1)
        return dx, dw, None


_orig = F.conv3d
_mode = None


def _patched(x, w, b=None, stride=1, padding=0, *a, **k):
    if _mode and w.shape[2] == 3 and w.shape[1] > 1:
        y = _WinoFn.apply(x, w, _mode)
        return y if b is None else y + b.view(1, -1, 1, 1, 1)
    return _orig(x, w, b, stride, padding, *a, **k)


def main():
    global _mode
    ap = argparse.ArgumentParser()
    ap.add_argument("names", nargs="+")
    ap.add_argument("--layers", default="fwd,dgrad")
    ap.add_argument("--threads", type=int, default=8)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    # self-check of the algebra in fp64
    x, w = torch.randn(2, 3, 7, 6, 5, dtype=torch.float64), torch.randn(4, 3, 3, 3, 3, dtype=torch.float64)
    assert (winograd_conv3d(x, w) - _orig(x, w, None, 1, 1)).abs().max() < 1e-12
    F.conv3d = _patched
    for name in a.names:
        g = Golden(name)
        for label, mode in (("direct", None), ("winograd", a.layers)):
            _mode = mode
            r = run_oracle(g, dtype=torch.float32, train=True)
            out = []
            for k, v in r["outs"].items():
                ref = g[f"f64/train/{k}"]
                out.append(f"{k} {np.abs(v.detach().double().numpy() - ref).max():.2e}")
            out.append(f"loss {abs(r['loss'].item() - float(g['f64/train/loss'])):.2e}")
            worst = 0.0
            for k, gr in r["grads"].items():
                ref = g[f"f64/grad/{k}"]
                from _golden import gprobe, zero_grad_keys
                if k in zero_grad_keys(g.spec, g.model):
                    continue
                got = gprobe(gr)
                worst = max(worst, np.abs(got[3:] - ref[3:]).max() / max(ref[2], 1e-30))
            out.append(f"worst grad probe {worst:.2e}")
            print(f"{name:18s} {label:9s} " + "  ".join(out), flush=True)
    F.conv3d = _orig


if __name__ == "__main__":
    main()
