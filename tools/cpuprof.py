#!/usr/bin/env python3
"""Host-side cost of issuing one train step (cProfile over 30 steps, no synchronisation inside)."""
import cProfile, os, pstats, sys, time
import torch
from torch import nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import transmf_ad_amd as T
if len(sys.argv) > 1:
    T.set_conv_precision(sys.argv[1])
dev = "cuda:0"
net = T.model_ad(128, 3, 4, 32, 512, 0.).to(dev).train()
opt = torch.optim.Adam(net.parameters(), lr=1e-4, fused=True)
B, S = 8, 96
mri = torch.rand((B, 1, S, S, S), device=dev); pet = torch.rand((B, 1, S, S, S), device=dev)
y = (torch.arange(B, device=dev) % 2).long(); ones = torch.ones_like(y); zeros = torch.zeros_like(y)
ce = nn.CrossEntropyLoss()


def step():
    opt.zero_grad()
    lo, dm, dp = net(mri, pet)
    ((ce(dm, ones) + ce(dp, zeros)) / 2 + ce(lo, y)).backward()
    opt.step()


for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"issue time {1e3 * (t1 - t0) / 10:.2f} ms/step, drained after {1e3 * (t2 - t0) / 10:.2f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(30): step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
