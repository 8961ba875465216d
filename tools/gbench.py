#!/usr/bin/env python3
"""A/B: eager step vs step with the fusion transformer's forward/backward replayed from hipGraphs
(torch.cuda.make_graphed_callables)."""
import os, sys, time
import torch
from torch import nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import transmf_ad_amd as T
dev = "cuda:0"
B, S = 8, 96
mri = torch.rand((B, 1, S, S, S), device=dev); pet = torch.rand((B, 1, S, S, S), device=dev)
y = (torch.arange(B, device=dev) % 2).long()
ce = nn.CrossEntropyLoss()
def make(graphed):
    torch.manual_seed(0)
    net = T.model_ad(128, 3, 4, 32, 512, 0.).to(dev).train()
    if graphed:
        tm = torch.randn((B, 216, 128), device=dev, requires_grad=True)
        tp = torch.randn((B, 216, 128), device=dev, requires_grad=True)
        net.fuse_transformer = torch.cuda.make_graphed_callables(net.fuse_transformer, (tm, tp))
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    def step():
        opt.zero_grad()
        lo, dm, dp = net(mri, pet)
        loss = (ce(dm, torch.ones_like(y)) + ce(dp, torch.zeros_like(y))) / 2 + ce(lo, y)
        loss.backward(); opt.step(); return loss
    return step
def wall(fn, reps=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
e = make(False); g = make(True)
for _ in range(2):
    print(f"eager {wall(e):.3f} ms   graphed-fusion {wall(g):.3f} ms", flush=True)
