#!/usr/bin/env python3
"""Run the two-stream bit-reproducibility check of the full-size train step N times and count failures.
   TMF_LIB=<other build> python tools/repro_loop.py [--n 12] [--mode fp32|bf16]"""
import argparse
import os
import sys

import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import transmf_ad_amd as T      # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=12)
ap.add_argument("--mode", default="fp32")
ap.add_argument("--one-call", type=int, default=1)
a = ap.parse_args()
from transmf_ad_amd import ops     # noqa: E402
ops.SNET_ONE_CALL = bool(a.one_call)
if a.mode == "bf16":
    T.set_conv_precision("bf16"); T.set_activation_storage("bf16")
torch.manual_seed(0)
net = T.model_ad(128, 3, 4, 32, 512, 0.).to("cuda:0")
B, S = 8, 96
g = torch.Generator(device="cuda:0").manual_seed(1)
mri = torch.rand((B, 1, S, S, S), device="cuda:0", generator=g)
pet = torch.rand((B, 1, S, S, S), device="cuda:0", generator=g)
y = (torch.arange(B, device="cuda:0") % 2).long()
ce = nn.CrossEntropyLoss()
ref, bad_runs, bad_names = None, 0, {}
for it in range(a.n + 1):
    torch.manual_seed(123)
    net.train()
    net.zero_grad(set_to_none=True)
    lo, dm, dp = net(mri, pet)
    loss = (ce(dm, torch.ones_like(y)) + ce(dp, torch.zeros_like(y))) / 2 + ce(lo, y)
    loss.backward()
    torch.cuda.synchronize()
    cur = {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
    if ref is None:
        ref = cur
        continue
    bad = [n for n in ref if not torch.equal(cur[n], ref[n])]
    if bad:
        bad_runs += 1
        for n in bad:
            bad_names[n] = bad_names.get(n, 0) + 1
print(f"lib={os.environ.get('TMF_LIB', 'default')} mode={a.mode} one_call={a.one_call}: {bad_runs} of {a.n} runs differ from the first; "
      f"{sorted(bad_names.items(), key=lambda kv: -kv[1])[:6]}")
