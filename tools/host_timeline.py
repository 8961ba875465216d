#!/usr/bin/env python3
"""Where is the HOST while the GPU runs the train step?  The bench's step (model_ad, batch 8, 96^3, fp32 Winograd mode) with
time.perf_counter() marks between its statements — no extra synchronisation: the two loss.item() calls of the reference's step
are the only points where the host waits.  Output: mean offset of every mark into the step (ms) over the timed steps and the
mean step time.  A mark that comes late relative to the kernel trace of the same step (tools/trace_seq.py) means the GPU
waits for the host there.

    python tools/host_timeline.py [--steps 20] [--setup 40]
"""
import argparse
import sys
import time

import numpy as np
import torch
import torch.nn as nn

sys.path.insert(0, ".")
from transmf_ad_amd import model_ad, ops, _lib            # noqa: E402
from transmf_ad_amd.optim import Adam                     # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--setup", type=int, default=40)
ap.add_argument("--B", type=int, default=8)
ap.add_argument("--S", type=int, default=96)
ap.add_argument("--ddp", action="store_true", help="wrap the model in GradAllReduce over a 1-rank RCCL group (the N > 1 machinery on one GPU)")
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = model_ad(dim=128, depth=3, heads=4, dim_head=32, mlp_dim=512, dropout=0.0).to(dev)
if a.ddp:
    import os
    import torch.distributed as dist
    from transmf_ad_amd.parallel import GradAllReduce
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    net = GradAllReduce(net)
opt = Adam(net.parameters(), lr=1e-4)
crit = nn.CrossEntropyLoss()
rs = np.random.RandomState(0)
mri = torch.from_numpy(rs.rand(a.B, 1, a.S, a.S, a.S).astype(np.float32)).to(dev)
pet = torch.from_numpy(rs.rand(a.B, 1, a.S, a.S, a.S).astype(np.float32)).to(dev)
label = (torch.arange(a.B, device=dev) % 2).long()
ones = torch.ones(a.B, dtype=torch.int64, device=dev)
zeros = torch.zeros(a.B, dtype=torch.int64, device=dev)
names = ["zero_grad", "forward returned", "3 x criterion", "ce_loss.item()", "ad_loss.item()", "loss = ad + ce", "backward returned", "opt.step returned"]


def step(marks):
    net.train()
    t0 = time.perf_counter()
    opt.zero_grad(); marks.append(time.perf_counter() - t0)
    lo, dm, dp = net(mri, pet); marks.append(time.perf_counter() - t0)
    ce = crit(lo, label)
    ad = (crit(dm, ones) + crit(dp, zeros)) / 2; marks.append(time.perf_counter() - t0)
    ce.item(); marks.append(time.perf_counter() - t0)
    ad.item(); marks.append(time.perf_counter() - t0)
    loss = ad + ce; marks.append(time.perf_counter() - t0)
    loss.backward(); marks.append(time.perf_counter() - t0)
    opt.step(); marks.append(time.perf_counter() - t0)


for _ in range(a.setup):
    step([])
torch.cuda.synchronize()
import gc
gc.collect(); gc.disable()
rows = []
t_all = time.perf_counter()
for _ in range(a.steps):
    m = []
    step(m)
    rows.append(m)
torch.cuda.synchronize()
t_all = time.perf_counter() - t_all
rows = np.array(rows) * 1e3
print(f"step {t_all / a.steps * 1e3:.3f} ms (host loop incl. final sync); host marks, ms into the step (mean / min / max):")
prev = 0.0
for i, n in enumerate(names):
    print(f"  {n:22s} {rows[:, i].mean():7.3f}  {rows[:, i].min():7.3f}  {rows[:, i].max():7.3f}   (+{rows[:, i].mean() - prev:6.3f})")
    prev = rows[:, i].mean()
