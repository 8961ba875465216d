#!/usr/bin/env python3
"""Where the N>1 gradient machinery spends time, measured on ONE GPU (1-rank RCCL group, TMF_DDP_FORCE=1):
   variants = hooks only / + pack / + all-reduce (full)."""
import os, sys, time
os.environ["TMF_DDP_FORCE"] = "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
import torch, torch.distributed as dist
from torch import nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import transmf_ad_amd as T
from transmf_ad_amd import parallel

dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
B, S = 8, 96
mri = torch.rand((B, 1, S, S, S), device=dev); pet = torch.rand((B, 1, S, S, S), device=dev)
y = (torch.arange(B, device=dev) % 2).long(); ones = torch.ones_like(y); zeros = torch.zeros_like(y)
ce = nn.CrossEntropyLoss()


def run(tag, wrap, patch=None):
    torch.manual_seed(0)
    net = T.model_ad(128, 3, 4, 32, 512, 0.).to(dev)
    if wrap:
        net = parallel.GradAllReduce(net)
        if patch:
            patch(net)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, fused=True)

    def step():
        net.train(); opt.zero_grad()
        lo, dm, dp = net(mri, pet)
        ((ce(dm, ones) + ce(dp, zeros)) / 2 + ce(lo, y)).backward()
        opt.step()
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize()
    print(f"{tag:40s} {(time.perf_counter() - t0) / 20 * 1e3:7.3f} ms/step", flush=True)


def hooks_only(net):
    def launch(b):
        class W:
            def wait(self): pass
        b.work = W()
    net._launch = launch


def no_allreduce(net):
    orig_views = net._views
    def launch(b):
        views = orig_views(b)
        grads = [p.grad.reshape(p.shape) for p in b.params if p.grad is not None]
        dst = [v for p, v in zip(b.params, views) if p.grad is not None]
        torch._foreach_copy_(dst, grads)
        class W:
            def wait(self): pass
        b.work = W()
    net._launch = launch


run("bare module", False)
run("hooks + finalize only", True, hooks_only)
run("hooks + pack (no all-reduce)", True, no_allreduce)
run("full (1-rank RCCL all-reduce)", True)
run("bare module again", False)
dist.destroy_process_group()
