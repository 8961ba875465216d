#!/usr/bin/env python3
"""CPU experiment: would a bf16 Winograd kernel stay inside the bands of the bf16 modes?

The oracle's conv_mode "bf16" (oracle/tmf_oracle.py: every 3x3x3 product takes its two operands rounded to bf16, exact
accumulation) is run beside a variant "bf16w" in which the Cin > 1 3x3x3 layers take the Winograd form F(2x2x2, 3x3x3) the
way a v_mfma_f32_32x32x16_bf16 kernel would have to run it: the transforms in fp32, the TRANSFORMED operands
    V = B^T d B (input tile),  U = G g G^T (filter),  Z = A dy A^T (output-gradient tile, weight gradient)
rounded to bf16 — they are the matrix instruction's operands —, products accumulated in the run's dtype, output transform in
fp32.  Forward, data gradient (the same form on dz with the flipped filter) and weight gradient (dU_p = V_p^T Z_p,
dw = G^T dU G) all take it; the first block (Cin = 1) and the 1x1x1 layer stay as in "bf16".

    python tools/winograd_bf16_numerics.py --case ad_128_b8           one train step against the fixture's fp32 golden values:
                                                                       the columns of tests/test_gpu_model.py CFG3_TOL
    python tools/winograd_bf16_numerics.py --trajectory 30            30 Adam steps (lr 1e-4, batch 8, 48^3 structured volumes,
                                                                       fixed masks: test_bf16_training_follows_the_fp32_trajectory)
                                                                       in fp32 / bf16 / bf16w from one start
"""
import argparse
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
from oracle import tmf_oracle as O          # noqa: E402
from oracle import params as P              # noqa: E402

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def _rb(t):
    return t.float().bfloat16().to(t.dtype)


def _t3(m, t, first):
    """apply the matrix m along the three consecutive axes first, first + 1, first + 2 of t"""
    for ax in range(first, first + 3):
        t = torch.tensordot(t, m.to(t.dtype), dims=([ax], [1])).movedim(-1, ax)
    return t


def _tiles(xp):
    """padded sample [C, D + 2, H + 2, W + 2] (even D, H, W) -> [C, D/2, H/2, W/2, 4, 4, 4] (tiles 2 apart)"""
    return xp.unfold(1, 4, 2).unfold(2, 4, 2).unfold(3, 4, 2)


def _pad_even(x):
    B, C, D, H, W = x.shape
    return F.pad(x, (1, 1 + (W & 1), 1, 1 + (H & 1), 1, 1 + (D & 1)))


def wino_conv(x, w, rnd):
    """y = conv3d(x, w, padding 1) in the Winograd form; rnd is applied to the transformed operands V and U."""
    B, C, D, H, W = x.shape
    O_ = w.shape[0]
    u = rnd(_t3(G, w, 2)).permute(2, 3, 4, 1, 0).reshape(64, C, O_)
    xp = _pad_even(x)
    out = []
    for b in range(B):                                             # per sample: the tile tensor is 8x the input
        v = rnd(_t3(BT, _tiles(xp[b]), 4))                         # [C, d, h, w, 4, 4, 4]
        d2, h2, w2 = v.shape[1:4]
        m = torch.bmm(v.permute(4, 5, 6, 1, 2, 3, 0).reshape(64, -1, C), u)            # [64, tiles, O]
        y = _t3(AT, m.reshape(4, 4, 4, d2, h2, w2, O_), 0)         # [2, 2, 2, d, h, w, O]
        out.append(y.permute(6, 3, 0, 4, 1, 5, 2).reshape(O_, 2 * d2, 2 * h2, 2 * w2)[:, :D, :H, :W])
    return torch.stack(out)


def wino_wgrad(x, dz, rnd):
    """dw [O, C, 3, 3, 3] of conv3d(x, w, padding 1): dU_p = sum over tiles V_p^T Z_p, dw = G^T dU G; rnd on V and Z."""
    B, C, D, H, W = x.shape
    O_ = dz.shape[1]
    xp = _pad_even(x)
    dzp = F.pad(dz, (0, W & 1, 0, H & 1, 0, D & 1))
    du = torch.zeros((64, C, O_), dtype=x.dtype)
    for b in range(B):
        v = rnd(_t3(BT, _tiles(xp[b]), 4))
        d2, h2, w2 = v.shape[1:4]
        t = dzp[b].unfold(1, 2, 2).unfold(2, 2, 2).unfold(3, 2, 2)                     # [O, d, h, w, 2, 2, 2]
        z = rnd(_t3(AT.t().contiguous(), t, 4))                                         # A dy A^T: [O, d, h, w, 4, 4, 4]
        du += torch.bmm(v.permute(4, 5, 6, 0, 1, 2, 3).reshape(64, C, -1), z.permute(4, 5, 6, 1, 2, 3, 0).reshape(64, -1, O_))
    return _t3(G.t().contiguous(), du.reshape(4, 4, 4, C, O_), 0).permute(4, 3, 0, 1, 2).contiguous()


class ConvBf16Wino(torch.autograd.Function):
    """stand-in for oracle.tmf_oracle._ConvBf16: Winograd with bf16 operands for Cin > 1, the direct bf16 product otherwise"""
    @staticmethod
    def forward(ctx, x, w, pad):
        ctx.wino = pad == 1 and w.shape[1] > 1 and w.shape[2] == 3
        ctx.pad = pad
        if not ctx.wino:
            xr, wr = _rb(x), _rb(w)
            ctx.save_for_backward(xr, wr)
            return F.conv3d(xr, wr, None, stride=1, padding=pad)
        ctx.save_for_backward(x, w)
        return wino_conv(x, w, _rb)

    @staticmethod
    def backward(ctx, dz):
        x, w = ctx.saved_tensors
        if not ctx.wino:
            dzr = _rb(dz)
            dx = torch.nn.grad.conv3d_input(x.shape, w, dzr, stride=1, padding=ctx.pad) if ctx.needs_input_grad[0] else None
            return dx, torch.nn.grad.conv3d_weight(x, w.shape, dzr, stride=1, padding=ctx.pad), None
        dx = wino_conv(dz, w.flip(2, 3, 4).transpose(0, 1).contiguous(), _rb) if ctx.needs_input_grad[0] else None
        return dx, wino_wgrad(x, dz, _rb), None


def self_check():
    torch.manual_seed(1)
    x, w = torch.randn(2, 3, 7, 6, 5, dtype=torch.float64), torch.randn(4, 3, 3, 3, 3, dtype=torch.float64)
    dz = torch.randn(2, 4, 7, 6, 5, dtype=torch.float64)
    ident = lambda t: t
    assert (wino_conv(x, w, ident) - F.conv3d(x, w, None, 1, 1)).abs().max() < 1e-12
    assert (wino_wgrad(x, dz, ident) - torch.nn.grad.conv3d_weight(x, w.shape, dz, stride=1, padding=1)).abs().max() < 1e-11
    dx = wino_conv(dz, w.flip(2, 3, 4).transpose(0, 1).contiguous(), ident)
    assert (dx - torch.nn.grad.conv3d_input(x.shape, w, dz, stride=1, padding=1)).abs().max() < 1e-11


def set_mode(mode):
    """-> the oracle's conv_mode for `mode`; "bf16w" swaps the conv function of the bf16 restatement"""
    O._ConvBf16 = ConvBf16Wino if mode == "bf16w" else ORIG
    return "exact" if mode == "fp32" else "bf16"


def one_step(case, modes, dtype):
    from _golden import Golden, gprobe, probe, zero_grad_keys
    g = Golden(case)
    mri, pet, y = g.inputs()
    mri, pet, y = torch.from_numpy(mri).to(dtype), torch.from_numpy(pet).to(dtype), torch.from_numpy(y)
    k1, k2 = g.masks()
    zk = zero_grad_keys(g.spec, g.model)
    print(f"# {case}: batch {g.batch}, {g.size}; errors against the fixture's fp32 golden run (the bands of CFG3_TOL: logits 0.3, D 3e-2, "
          f"loss 6e-2, act 3e-2, conv-side |grad|-sum 0.6, fusion + heads |grad|-sum 0.35)")
    for mode in modes:
        t0 = time.time()
        S = O.to_state(g.arrays(), g.spec, dtype=dtype, requires_grad=True)
        probes = {}
        lo, dm, dp = O.model_ad_forward(S, mri, pet, dim=g.kw["dim"], depth=g.kw["depth"], heads=g.kw["heads"], train=True,
                                        dropout_masks=(torch.from_numpy(k1), torch.from_numpy(k2)), probes=probes, conv_mode=set_mode(mode))
        loss = O.adversarial_loss(lo, dm, dp, y)
        loss.backward()
        grads = O.grads_of(S, g.spec)
        e_lo = np.abs(lo.detach().double().numpy() - g["f32/train/logits"]).max()
        e_d = max(np.abs(dm.detach().double().numpy() - g["f32/train/d_mri"]).max(), np.abs(dp.detach().double().numpy() - g["f32/train/d_pet"]).max())
        e_loss = abs(loss.item() - float(g["f32/train/loss"]))
        e_act = 0.0
        for k in ("cls", "mri_cnn.conv4.3", "pet_cnn.conv4.3"):
            if k in probes and g.has(f"f32/probe/{k}"):
                ref = g[f"f32/probe/{k}"]
                e_act = max(e_act, np.abs(probe(probes[k]) - ref).max() / max(1.0, np.abs(ref).max()))
        gc = gt = pc = pt = 0.0
        for k, gr in grads.items():
            if k in zk:
                continue
            ref, got = g[f"f32/grad/{k}"], gprobe(gr)
            s_err = abs(got[1] - ref[1]) / max(ref[1], 1e-30)
            p_err = np.abs(got[3:] - ref[3:]).max() / max(ref[2], 1e-30)
            if "_cnn." in k:
                gc, pc = max(gc, s_err), max(pc, p_err)
            else:
                gt, pt = max(gt, s_err), max(pt, p_err)
        print(f"{mode:6s} logits {e_lo:.3e}  D {e_d:.3e}  loss {e_loss:.3e}  act {e_act:.3e}  |grad|-sum conv {gc:.3f} fusion+heads {gt:.3f}"
              f"  sampled grad elements / max: conv {pc:.3f} fusion+heads {pt:.3f}   ({time.time() - t0:.0f} s)", flush=True)
    O._ConvBf16 = ORIG


def trajectory(steps, modes, dtype, size):
    kw = dict(dim=128, depth=3, heads=4, dim_head=32, mlp_dim=512)
    spec = O.state_spec("model_ad", **kw)
    arrs = P.init_arrays(spec, seed=11)
    B = 8
    mri, pet, y = P.make_inputs(B, (size, size, size), seed=77, kind="blobs")
    k1, k2 = P.make_masks(B, seed=5)
    m, p_, yy = torch.from_numpy(mri).to(dtype), torch.from_numpy(pet).to(dtype), torch.from_numpy(y)
    masks = (torch.from_numpy(k1), torch.from_numpy(k2))
    res = {}
    print(f"# {steps} Adam steps (lr 1e-4), batch {B}, {size}^3 structured volumes, fixed masks, one start; the measured GPU numbers of the "
          f"bf16 modes: max |d loss| 0.029, max |d CE| 0.006 (bands 0.09 / 0.02)")
    for mode in modes:
        t0 = time.time()
        S = O.to_state(arrs, spec, dtype=dtype)
        prm = [S[k] for k, (kind, _s) in spec.items() if kind == "param"]
        opt = torch.optim.Adam(prm, lr=1e-4)
        cm = set_mode(mode)
        losses, ces = [], []
        for _ in range(steps):
            opt.zero_grad()
            lo, dm, dp = O.model_ad_forward(S, m, p_, train=True, dropout_masks=masks, conv_mode=cm, **{k: kw[k] for k in ("dim", "depth", "heads")})
            ce = F.cross_entropy(lo, yy)
            loss = O.adversarial_loss(lo, dm, dp, yy)
            loss.backward()
            opt.step()
            losses.append(loss.item()); ces.append(ce.item())
        res[mode] = (np.array(losses), np.array(ces))
        msg = f"{mode:6s} loss {losses[0]:.4f} -> {losses[-1]:.4f}  CE {ces[0]:.4f} -> {ces[-1]:.4f}"
        if mode != "fp32" and "fp32" in res:
            msg += f"   max |d loss| vs fp32 {np.abs(res[mode][0] - res['fp32'][0]).max():.4f}  max |d CE| {np.abs(res[mode][1] - res['fp32'][1]).max():.4f}"
        print(msg + f"   ({time.time() - t0:.0f} s)", flush=True)
    O._ConvBf16 = ORIG


ORIG = O._ConvBf16

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="")
    ap.add_argument("--trajectory", type=int, default=0)
    ap.add_argument("--size", type=int, default=48)
    ap.add_argument("--modes", default="fp32,bf16,bf16w")
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--f64", action="store_true", help="accumulate in fp64 (the tests' choice for the bf16 oracle) instead of fp32")
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    self_check()
    dt = torch.float64 if a.f64 else torch.float32
    if a.case:
        one_step(a.case, a.modes.split(","), dt)
    if a.trajectory:
        trajectory(a.trajectory, a.modes.split(","), dt, a.size)
