#!/usr/bin/env python3
"""Host-side time of the pieces of a train step (python-level wall time per call, GPU work asynchronous):
where the training thread spends its time between the reference's mid-step sync and the end of backward."""
import os
import sys
import time
import collections

import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import transmf_ad_amd as T          # noqa: E402
from transmf_ad_amd import ops       # noqa: E402

acc = collections.defaultdict(float)
cnt = collections.defaultdict(int)


def wrap(cls, name):
    orig = getattr(cls, name)

    def timed(*a, **k):
        t0 = time.perf_counter()
        r = orig(*a, **k)
        acc[f"{cls.__name__}.{name}"] += time.perf_counter() - t0
        cnt[f"{cls.__name__}.{name}"] += 1
        return r
    setattr(cls, name, staticmethod(timed))


for c in (ops.SNetTrain, ops.FusionTrain):
    wrap(c, "forward")
    wrap(c, "backward")
dev = "cuda:0"
torch.manual_seed(0)
net = T.model_ad(128, 3, 4, 32, 512, 0.).to(dev)
opt = torch.optim.Adam(net.parameters(), lr=1e-4, fused=True)
B, S = 8, 96
mri = torch.rand((B, 1, S, S, S), device=dev)
pet = torch.rand((B, 1, S, S, S), device=dev)
y = (torch.arange(B, device=dev) % 2).long()
ones, zeros = torch.ones_like(y), torch.zeros_like(y)
crit = nn.CrossEntropyLoss()
seg = collections.defaultdict(float)
N = 30
for it in range(N + 10):
    if it == 10:
        acc.clear(); cnt.clear(); seg.clear()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    net.train(); opt.zero_grad()
    t1 = time.perf_counter()
    lo, dm, dp = net(mri, pet)
    ce = crit(lo, y); ad = (crit(dm, ones) + crit(dp, zeros)) / 2
    t2 = time.perf_counter()
    ce.item(); ad.item()
    t3 = time.perf_counter()
    loss = ad + ce
    loss.backward()
    t4 = time.perf_counter()
    opt.step()
    t5 = time.perf_counter()
    seg["train()+zero_grad"] += t1 - t0; seg["forward issue (incl. heads, loss)"] += t2 - t1
    seg["2x .item() (waits for the GPU forward)"] += t3 - t2; seg["backward issue"] += t4 - t3; seg["opt.step issue"] += t5 - t4
torch.cuda.synchronize()
for k, v in seg.items():
    print(f"{k:44s} {v / N * 1e3:8.3f} ms per step")
for k, v in acc.items():
    print(f"   {k:41s} {v / N * 1e3:8.3f} ms per step  ({cnt[k] // N} calls)")
