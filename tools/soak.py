#!/usr/bin/env python3
"""Soak: N train steps on one fixed batch per precision mode — loss must fall, stay finite, memory must not grow."""
import os, sys
import torch
from torch import nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import transmf_ad_amd as T
dev = "cuda:0"
B, S, N = 8, 96, int(os.environ.get("SOAK_STEPS", "150"))
for prec, store in (("fp32", "fp32"), ("fp32x", "fp32"), ("bf16", "fp32"), ("bf16", "bf16")):
    T.set_conv_precision(prec); T.set_activation_storage(store)
    torch.manual_seed(0)
    net = T.model_ad(128, 3, 4, 32, 512, 0.).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, fused=True)
    g = torch.Generator(device=dev).manual_seed(1)
    mri = torch.rand((B, 1, S, S, S), device=dev, generator=g); pet = torch.rand((B, 1, S, S, S), device=dev, generator=g)
    y = (torch.arange(B, device=dev) % 2).long(); ones = torch.ones_like(y); zeros = torch.zeros_like(y)
    ce = nn.CrossEntropyLoss()
    losses, mem = [], []
    for i in range(N):
        net.train(); opt.zero_grad()
        lo, dm, dp = net(mri, pet)
        loss = (ce(dm, ones) + ce(dp, zeros)) / 2 + ce(lo, y)
        loss.backward(); opt.step()
        if i % 10 == 0 or i == N - 1:
            losses.append(loss.item()); mem.append(torch.cuda.memory_allocated() >> 20)
    ok = all(l == l for l in losses) and losses[-1] < losses[0] and max(mem[2:]) - min(mem[2:]) < 64
    print(f"{prec:6s} storage={store}: loss {losses[0]:.4f} -> {losses[-1]:.4f} (cls head only: {ce(lo, y).item():.4f}), "
          f"allocated {mem[2]}..{mem[-1]} MiB, peak {torch.cuda.max_memory_allocated() >> 20} MiB  {'OK' if ok else 'FAIL'}", flush=True)
    del net, opt
    torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
T.set_conv_precision("fp32"); T.set_activation_storage("fp32")
