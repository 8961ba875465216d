#!/usr/bin/env python3
"""Time the four passes of the fused first block (conv1_fused.hip), fp32 and bf16 products, per launch.

    python tools/c1_time.py [--S 128] [--reps 20]        (TMF_C1_BLOCKS / TMF_C1_FWD_MULT are read by the library)
"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transmf_ad_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--S", type=int, default=128)
ap.add_argument("--reps", type=int, default=20)
args = ap.parse_args()
B, S, C = 8, args.S, 32
dev = "cuda:0"
x = torch.rand((B, S, S, S), device=dev)
w = torch.randn((27, C), device=dev) * 0.2
sc, sh = torch.ones(C, device=dev), torch.zeros(C, device=dev)
mu, isd = torch.zeros(C, device=dev), torch.ones(C, device=dev)
coef = torch.zeros((2, C), device=dev)
out32 = torch.empty((B, S // 2, S // 2, S // 2, C), device=dev)
out16 = torch.empty((B, S // 2, S // 2, S // 2, C), device=dev, dtype=torch.bfloat16)
nb = _lib.query("tmf_c1_blocks", B, S, S, S, C)
part = torch.empty((nb, 2, C), device=dev)
dw = torch.empty((27, C), device=dev)
nby = _lib.query("tmf_c1_bwd_wgrad_workspace_bytes", B, S, S, S, C)
ws = torch.empty((nby // 4,), device=dev)
st = torch.cuda.current_stream().cuda_stream
p = lambda t: t.data_ptr()
passes = {
    "f32 stats": lambda: _lib.call("tmf_c1_stats", p(x), p(w), p(part), B, S, S, S, C, st),
    "f32 fwd": lambda: _lib.call("tmf_c1_bn_pool_fwd", p(x), p(w), p(sc), p(sh), p(out32), B, S, S, S, C, 0.01, st),
    "f32 reduce": lambda: _lib.call("tmf_c1_bwd_reduce", p(x), p(w), p(sc), p(sh), p(mu), p(isd), p(out32), p(part), B, S, S, S, C,
                                    0.01, st),
    "f32 wgrad": lambda: _lib.call("tmf_c1_bwd_wgrad", p(x), p(w), p(sc), p(sh), p(mu), p(isd), p(coef), p(out32), p(dw), p(ws), nby,
                                   B, S, S, S, C, 0.01, 0, st),
    "b16 stats": lambda: _lib.call("tmf_c1_stats_bf16", p(x), p(w), p(part), B, S, S, S, C, st),
    "b16 fwd": lambda: _lib.call("tmf_c1_bn_pool_fwd_bf16", p(x), p(w), p(sc), p(sh), p(out16), B, S, S, S, C, 0.01, 1, st),
    "b16 reduce": lambda: _lib.call("tmf_c1_bwd_reduce_bf16", p(x), p(w), p(sc), p(sh), p(mu), p(isd), p(out16), p(part), B, S, S, S,
                                    C, 0.01, 1, st),
    "b16 wgrad": lambda: _lib.call("tmf_c1_bwd_wgrad_bf16", p(x), p(w), p(sc), p(sh), p(mu), p(isd), p(coef), p(out16), p(dw), p(ws),
                                   nby, B, S, S, S, C, 0.01, 1, 0, st),
}
res = []
for name, fn in passes.items():
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    res.append((name, e0.elapsed_time(e1) / args.reps * 1e3))
print(f"S={S} blocks={nb} C1_BLOCKS={os.environ.get('TMF_C1_BLOCKS', '-')} FWD_MULT={os.environ.get('TMF_C1_FWD_MULT', '-')}: " +
      "  ".join(f"{n} {t:.1f}" for n, t in res) + "  (us per launch)")
