#!/usr/bin/env python3
"""Inference (val_step) throughput: eval-mode forward under no_grad, single-pass blocks vs the two-pass form."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import transmf_ad_amd as T
from transmf_ad_amd import ops
dev = "cuda:0"
B, S = 8, 96
net = T.model_ad(128, 3, 4, 32, 512, 0.).to(dev).eval()
mri = torch.rand((B, 1, S, S, S), device=dev); pet = torch.rand((B, 1, S, S, S), device=dev)


def wall(reps=20, warm=5):
    with torch.no_grad():
        for _ in range(warm): net(mri, pet)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): net(mri, pet)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for fused in (True, False, True):
    ops.FUSE_EVAL_BLOCKS = fused
    t = wall()
    print(f"eval forward, B={B}, {S}^3, single-pass blocks={fused}: {t:.3f} ms  ({B / t * 1e3:.0f} pairs/s)", flush=True)
