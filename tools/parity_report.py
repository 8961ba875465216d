#!/usr/bin/env python3
"""Print the parity margins of the HIP path against the golden fixtures (MI355X only).

    python tools/parity_report.py [--cases a,b,...] [--modes fp32,bf16,bf16s,fp32x] [--grads]

Per fixture and mode: |d outputs|, |d loss|, |d cls| against the reference's fp32 (and fp64, where the fixture has it)
run; with --grads the per-parameter gradient probes (16 sampled elements relative to the reference tensor's max-abs;
ratio of the |.|-sums) with the worst parameters named.  The test tolerances in tests/test_gpu_model.py are set from
this table (numbers recorded in DESIGN.md §4).
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _golden import Golden, available, gprobe, probe, zero_grad_keys          # noqa: E402
from test_gpu_model import build, step                # noqa: E402

ALL = ["ad_tiny", "ad_ragged", "cnn_tiny", "single_mid", "cnn_mid", "ad_mid", "ad_full_b2", "ad_full_b2_blobs",
       "ad_adni_b2", "ad_full_b8", "ad_128_b8"]
ap = argparse.ArgumentParser()
ap.add_argument("--cases", default=",".join(ALL))
ap.add_argument("--modes", default="fp32")
ap.add_argument("--grads", action="store_true")
args = ap.parse_args()

import transmf_ad_amd as T              # noqa: E402

MODES = {"fp32": ("fp32", "fp32"), "bf16": ("bf16", "fp32"), "bf16s": ("bf16", "bf16"), "fp32x": ("fp32x", "fp32")}

for name in args.cases.split(","):
    if not available(name):
        continue
    g = Golden(name)
    for mode in args.modes.split(","):
        prec, store = MODES[mode]
        T.set_conv_precision(prec)
        T.set_activation_storage(store)
        try:
            net = build(g)
            seen = {}
            if g.model == "model_ad":
                net.fuse_transformer.register_forward_hook(lambda _m, _i, o: seen.__setitem__("cls", o))
            for c in ("mri_cnn", "pet_cnn", "cnn"):
                if hasattr(net, c):
                    getattr(net, c).register_forward_hook(
                        lambda _m, _i, o, c=c: seen.__setitem__(f"{c}.conv4.3", o.contiguous()))
            outs, loss = step(net, g, train=True)
        finally:
            T.set_conv_precision("fp32")
            T.set_activation_storage("fp32")
        msg = [f"{name:17s} {mode:6s}"]
        for p in ("f32", "f64"):
            if not g.has(f"{p}/train/logits"):
                continue
            d = {k: np.abs(v.detach().double().cpu().numpy() - g[f"{p}/train/{k}"]).max() for k, v in outs.items()}
            msg.append(f"vs {p}: logits {d['logits']:.2e} D {max(v for k, v in d.items() if k != 'logits') if len(d) > 1 else 0:.2e}"
                       f" loss {abs(loss.item() - float(g[f'{p}/train/loss'])):.2e}")
            for k, t in seen.items():
                if g.has(f"{p}/probe/{k}"):
                    ref = g[f"{p}/probe/{k}"]
                    msg.append(f"{k.split('.')[0]}: {np.abs(probe(t) - ref).max() / max(1.0, np.abs(ref).max()):.2e}")
        if g.has("f64/train/logits"):
            msg.append(f"(ref f32 vs f64: {np.abs(g['f32/train/logits'] - g['f64/train/logits']).max():.2e})")
        print("  ".join(msg), flush=True)
        if args.grads:
            p = "f64" if g.has("f64/grad/" + next(iter(dict(net.named_parameters())))) else "f32"
            zk = zero_grad_keys(g.spec, g.model)
            rows = []
            for k, prm in net.named_parameters():
                if k in zk:
                    continue
                ref = g[f"{p}/grad/{k}"]
                got = gprobe(prm.grad if prm.grad is not None else torch.zeros_like(prm))
                e_s = np.abs(got[3:] - ref[3:]).max() / max(ref[2], 1e-30)
                e_a = abs(got[1] - ref[1]) / max(ref[1], 1e-30)
                e_m = abs(got[2] - ref[2]) / max(ref[2], 1e-30)
                rows.append((e_s, e_a, e_m, k))
            rows.sort(reverse=True)
            grp = {"conv": [r for r in rows if "_cnn." in r[3] or r[3].startswith("cnn.")],
                   "fusion": [r for r in rows if r[3].startswith("fuse_transformer")],
                   "heads": [r for r in rows if r[3].startswith(("fc", "D."))]}
            for gn, rr in grp.items():
                if rr:
                    print(f"    grads vs ref {p} [{gn}]: worst sample/max {rr[0][0]:.2e} ({rr[0][3]}), "
                          f"worst |sum| ratio err {max(r[1] for r in rr):.2e}, worst max err {max(r[2] for r in rr):.2e}")
            if p == "f64" and g.has("f32/grad/" + rows[0][3]):
                own = max(np.abs(g[f"f32/grad/{k}"][3:] - g[f"f64/grad/{k}"][3:]).max() / max(g[f"f64/grad/{k}"][2], 1e-30)
                          for _a, _b, _c, k in rows)
                print(f"    (reference fp32 vs its fp64 grads, same measure: {own:.2e})")
        del net
        torch.cuda.empty_cache()
