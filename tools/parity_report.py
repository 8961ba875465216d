#!/usr/bin/env python3
"""Print the parity margins of the HIP path against every golden fixture (MI355X only)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _golden import Golden, available, probe          # noqa: E402
from test_gpu_model import build, step                # noqa: E402

for name in ["ad_tiny", "ad_ragged", "cnn_tiny", "single_mid", "cnn_mid", "ad_mid", "ad_full_b2", "ad_full_b8"]:
    if not available(name):
        continue
    g = Golden(name)
    net = build(g)
    seen = {}
    if g.model == "model_ad":
        net.fuse_transformer.register_forward_hook(lambda _m, _i, o: seen.__setitem__("cls", o))
    outs, loss = step(net, g, train=True)
    msg = [f"{name:11s}"]
    for prec in ("f32", "f64"):
        if not g.has(f"{prec}/train/logits"):
            continue
        d = max(np.abs(v.detach().double().cpu().numpy() - g[f"{prec}/train/{k}"]).max() for k, v in outs.items())
        msg.append(f"|d out| vs ref {prec}: {d:.2e}  |d loss|: {abs(loss.item() - float(g[f'{prec}/train/loss'])):.2e}")
        if "cls" in seen and g.has(f"{prec}/probe/cls"):
            msg.append(f"cls: {np.abs(probe(seen['cls']) - g[f'{prec}/probe/cls']).max():.2e}")
    if g.has("f64/train/logits"):
        msg.append(f"(ref f32 vs f64: {np.abs(g['f32/train/logits'] - g['f64/train/logits']).max():.2e})")
    print("  ".join(msg), flush=True)
