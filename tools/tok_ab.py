import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import transmf_ad_amd as T
from transmf_ad_amd import ops
dev="cuda:0"
torch.manual_seed(0)
net = T.model_ad(128, 3, 4, 32, 512, 0.).to(dev).train()
B=8
tok_m = torch.randn((B, 216, 128), device=dev, requires_grad=True)
tok_p = torch.randn((B, 216, 128), device=dev, requires_grad=True)
def wall(fn, reps=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/reps*1e3
def fusion():
    net.fuse_transformer.zero_grad()
    net.fuse_transformer(tok_m, tok_p).sum().backward()
def fusion_fwd():
    with torch.no_grad(): net.fuse_transformer(tok_m, tok_p)
for fused in (True, False, True):
    ops.FUSE_TOKEN_LINEARS = fused
    print("fused" if fused else "unfused", "fwd+bwd %.3f ms" % wall(fusion), " fwd %.3f ms" % wall(fusion_fwd), flush=True)
