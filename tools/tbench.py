#!/usr/bin/env python3
"""Wall-time split of one train step (HIP events around segments, B=8, 96^3)."""
import os, sys, time
import torch
from torch import nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import transmf_ad_amd as T

dev = "cuda:0"
torch.manual_seed(0)
net = T.model_ad(128, 3, 4, 32, 512, 0.).to(dev).train()
opt = torch.optim.Adam(net.parameters(), lr=1e-4)
B, S = 8, 96
mri = torch.rand((B, 1, S, S, S), device=dev)
pet = torch.rand((B, 1, S, S, S), device=dev)
y = (torch.arange(B, device=dev) % 2).long()
ce = nn.CrossEntropyLoss()


def wall(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def full():
    opt.zero_grad()
    lo, dm, dp = net(mri, pet)
    loss = (ce(dm, torch.ones_like(y)) + ce(dp, torch.zeros_like(y))) / 2 + ce(lo, y)
    loss.backward()
    opt.step()


def snets():
    net.zero_grad()
    a, b = T.mymodel._two_streams(net.mri_cnn, mri, net.pet_cnn, pet)
    (a.sum() + b.sum()).backward()


tok_m = torch.randn((B, 216, 128), device=dev, requires_grad=True)
tok_p = torch.randn((B, 216, 128), device=dev, requires_grad=True)


def fusion():
    net.fuse_transformer.zero_grad()
    net.fuse_transformer(tok_m, tok_p).sum().backward()


def fusion_fwd():
    with torch.no_grad():
        net.fuse_transformer(tok_m, tok_p)


cls = torch.randn((B, 512), device=dev, requires_grad=True)
vec = torch.randn((B, 128), device=dev, requires_grad=True)


def heads():
    lo = net.fc_cls(cls)
    dm, dp = net.D(vec), net.D(vec * 2)
    ((ce(dm, torch.ones_like(y)) + ce(dp, torch.zeros_like(y))) / 2 + ce(lo, y)).backward()


def adam():
    opt.step()


def zero():
    opt.zero_grad()


full()
for name, fn in [("full step", full), ("two sNets fwd+bwd", snets), ("fusion transformer fwd+bwd", fusion),
                 ("fusion transformer fwd", fusion_fwd), ("heads+loss fwd+bwd", heads), ("Adam step", adam),
                 ("zero_grad", zero)]:
    print(f"{name:32s} {wall(fn):8.3f} ms", flush=True)

# CPU issue time (no synchronisation inside the timed region)
def cpu_issue(fn_f, reps=10):
    ts_f, ts_b = [], []
    for _ in range(reps + 3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn_f()
        t1 = time.perf_counter()
        out.backward()
        t2 = time.perf_counter()
        ts_f.append(t1 - t0); ts_b.append(t2 - t1)
    return sum(ts_f[3:]) / reps * 1e3, sum(ts_b[3:]) / reps * 1e3


f, b = cpu_issue(lambda: net.fuse_transformer(tok_m, tok_p).sum())
print(f"fusion transformer CPU issue: fwd {f:.3f} ms, bwd {b:.3f} ms", flush=True)
one = net.fuse_transformer.layers[0][0]
f, b = cpu_issue(lambda: one(tok_m, context=tok_p).sum())
print(f"one Transformer instance CPU issue: fwd {f:.3f} ms, bwd {b:.3f} ms", flush=True)
ln = one.layers[0][0].norm
from transmf_ad_amd import ops
f, b = cpu_issue(lambda: ops.layer_norm(tok_m, ln.weight, ln.bias, 1e-5).sum())
print(f"ops.layer_norm CPU issue: fwd {f:.3f} ms, bwd {b:.3f} ms", flush=True)
f, b = cpu_issue(lambda: torch.nn.functional.layer_norm(tok_m, (128,), ln.weight, ln.bias, 1e-5).sum())
print(f"torch layer_norm CPU issue: fwd {f:.3f} ms, bwd {b:.3f} ms", flush=True)
q = torch.randn((B, 216, 128), device=dev, requires_grad=True)
kv = torch.randn((B, 216, 256), device=dev, requires_grad=True)
f, b = cpu_issue(lambda: ops.cross_attention(q, kv, 4, 32 ** -0.5).sum())
print(f"ops.cross_attention CPU issue: fwd {f:.3f} ms, bwd {b:.3f} ms", flush=True)
lin = one.layers[0][1].fn.net[0]
f, b = cpu_issue(lambda: lin(tok_m).sum())
print(f"nn.Linear CPU issue: fwd {f:.3f} ms, bwd {b:.3f} ms", flush=True)
