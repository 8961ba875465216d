#!/usr/bin/env python3
"""Per-parameter gradient error table (read-out loss of tests/test_gpu_model.py) for one fixture:
HIP path vs oracle fp64, next to oracle fp32 vs oracle fp64 (the inherent fp32 noise)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _golden import Golden, run_oracle          # noqa: E402
from test_gpu_model import build, DEV           # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "ad_mid"
g = Golden(name)
dim = g.kw["dim"]
rs = np.random.RandomState(3)
R1 = torch.from_numpy(rs.standard_normal((g.batch, 4 * dim)))
R2 = torch.from_numpy(rs.standard_normal((2, g.batch, dim)))


def readout(cls, m, p):
    return (cls * R1.to(cls)).sum() + (m.mean(dim=(2, 3, 4)) * R2[0].to(cls)).sum() + (p.mean(dim=(2, 3, 4)) * R2[1].to(cls)).sum()


G = {}
for dt in (torch.float64, torch.float32):
    r = run_oracle(g, dtype=dt, train=True, backward=False, keep_graph=True)
    P = r["probes"]
    readout(P["cls"], P["mri_cnn.conv4.3"], P["pet_cnn.conv4.3"]).backward()
    G[dt] = {k: r["state"][k].grad.double() for k, (kind, _s) in g.spec.items()
             if kind == "param" and r["state"][k].grad is not None}
net = build(g)
got = {}
for c in ("mri_cnn", "pet_cnn"):
    getattr(net, c).register_forward_hook(lambda _m, _i, o, c=c: got.__setitem__(c, o))
net.fuse_transformer.register_forward_hook(lambda _m, _i, o: got.__setitem__("cls", o))
mri, pet, _y = (torch.from_numpy(a).to(DEV) for a in g.inputs())
net.train()
net(mri, pet)
readout(got["cls"], got["mri_cnn"], got["pet_cnn"]).backward()
torch.cuda.synchronize()
print(f"{'param':58s} {'hip-vs-f64':>10s} {'f32-vs-f64':>10s} {'#elems>10%':>10s}")
for k, p in net.named_parameters():
    if k not in G[torch.float64] or p.grad is None:
        continue
    ref = G[torch.float64][k]
    mx = ref.abs().max().clamp_min(1e-30)
    d = (p.grad.double().cpu() - ref).abs()
    e1 = (d.max() / mx).item()
    e2 = ((G[torch.float32][k] - ref).abs().max() / mx).item()
    nbig = int((d > 0.1 * d.max()).sum()) if d.max() > 0 else 0
    flag = " <<<" if e1 > 3 * max(e2, 1e-5) else ""
    print(f"{k:58s} {e1:10.2e} {e2:10.2e} {nbig:10d}{flag}")
