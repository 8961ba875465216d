#!/usr/bin/env python3
"""How much would MORE concurrency buy the train step?  Two independent model_ad replicas (own parameters, own optimizer, own
pair of encoder streams) are stepped alternately from one thread WITHOUT host syncs, each on its own main stream, so the GPU sees
four encoder streams instead of two; aggregate pairs/s against one replica stepped the same way (bench.py --no-item-sync form).
   python tools/concurrency_probe.py [--steps 30]"""
import argparse, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import transmf_ad_amd as T

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--size", type=int, default=96)
a = ap.parse_args()
dev = torch.device("cuda:0")
crit = torch.nn.CrossEntropyLoss()


def make(seed):
    torch.manual_seed(seed)
    net = T.model_ad(dim=128, depth=3, heads=4, dim_head=32, mlp_dim=512, dropout=0.0).to(dev).train()
    opt = T.optim.Adam(net.parameters(), lr=1e-4)
    rs = np.random.RandomState(seed)
    mri = torch.from_numpy(rs.rand(a.batch, 1, a.size, a.size, a.size).astype(np.float32)).to(dev)
    pet = torch.from_numpy(rs.rand(a.batch, 1, a.size, a.size, a.size).astype(np.float32)).to(dev)
    y = torch.from_numpy(rs.randint(0, 2, a.batch)).to(dev)
    ones, zeros = torch.ones_like(y), torch.zeros_like(y)
    return net, opt, mri, pet, y, ones, zeros, torch.cuda.Stream(device=dev)


def step(r):
    net, opt, mri, pet, y, ones, zeros, st = r
    with torch.cuda.stream(st):
        opt.zero_grad()
        lo, dm, dp = net(mri, pet)
        loss = crit(lo, y) + (crit(dm, ones) + crit(dp, zeros)) / 2
        loss.backward()
        opt.step()


def run(reps):
    for _ in range(8):
        for r in reps:
            step(r)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        for r in reps:
            step(r)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return len(reps) * a.steps * a.batch / dt, dt / a.steps * 1e3


r1, r2 = make(1), make(2)
print("one replica : %.1f pairs/s (%.2f ms per step)" % run([r1]))
print("two replicas: %.1f pairs/s aggregate (%.2f ms per double step)" % run([r1, r2]))
print("one replica : %.1f pairs/s (%.2f ms per step)" % run([r2]))
