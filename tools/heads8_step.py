#!/usr/bin/env python3
"""Train step of model_ad (B = 8, 96^3, fp32) with both head geometries of the reference's scripts — 4 heads of 32
(kfold_train_adversarial.py:78-79) and 8 heads of 16 (train_adversarial.py:30-31) — on the fused per-instance transformer kernels and
on the one-launch-per-op path of the same entry (ops.FUSION_FUSED_KERNELS): volume-pairs/s, no host syncs inside the step."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import transmf_ad_amd as T              # noqa: E402
from transmf_ad_amd import ops          # noqa: E402

for heads in (4, 8):
    for fused in (True, False):
        ops.FUSION_FUSED_KERNELS = fused
        torch.manual_seed(0)
        net = T.model_ad(dim=128, depth=3, heads=heads, dim_head=128 // heads, mlp_dim=512, dropout=0.).cuda().train()
        opt = T.optim.Adam(net.parameters(), lr=1e-4)
        mri, pet = torch.rand(8, 1, 96, 96, 96, device="cuda"), torch.rand(8, 1, 96, 96, 96, device="cuda")
        y = torch.arange(8, device="cuda") % 2
        crit = torch.nn.CrossEntropyLoss()

        def step():
            opt.zero_grad()
            lo, dm, dp = net(mri, pet)
            loss = crit(lo, y) + (crit(dm, torch.ones_like(y)) + crit(dp, torch.zeros_like(y))) / 2
            loss.backward()
            opt.step()
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 30
        print(f"heads {heads} x {128 // heads}  fused kernels {str(fused):5s}: {8 / dt:7.1f} pairs/s  {dt * 1e3:.3f} ms per step", flush=True)
ops.FUSION_FUSED_KERNELS = True
