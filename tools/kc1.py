#!/usr/bin/env python3
"""Launch the four fused conv1 passes a few times (for rocprofv3 --pmc runs)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transmf_ad_amd import _lib
B, C = 8, 32
S = int(sys.argv[sys.argv.index("--S") + 1]) if "--S" in sys.argv else 96
BF = "--bf16" in sys.argv                  # the bf16-product kernels with bf16 pooled tensors
dev = "cuda:0"
x = torch.rand((B, S, S, S), device=dev)
w = torch.randn((27, C), device=dev) * 0.2
sc, sh = torch.ones(C, device=dev), torch.zeros(C, device=dev)
mu, isd = torch.zeros(C, device=dev), torch.ones(C, device=dev)
coef = torch.zeros((2, C), device=dev)
out = torch.empty((B, S // 2, S // 2, S // 2, C), device=dev, dtype=torch.bfloat16 if BF else torch.float32)
nb = _lib.query("tmf_c1_blocks", B, S, S, S, C)
part = torch.empty((nb, 2, C), device=dev)
dw = torch.empty((27, C), device=dev)
nby = _lib.query("tmf_c1_bwd_wgrad_workspace_bytes", B, S, S, S, C)
ws = torch.empty((nby // 4,), device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    if BF:
        _lib.call("tmf_c1_stats_bf16", x.data_ptr(), w.data_ptr(), part.data_ptr(), B, S, S, S, C, st)
        _lib.call("tmf_c1_bn_pool_fwd_bf16", x.data_ptr(), w.data_ptr(), sc.data_ptr(), sh.data_ptr(), out.data_ptr(), B, S, S, S, C,
                  0.01, 1, st)
        _lib.call("tmf_c1_bwd_reduce_bf16", x.data_ptr(), w.data_ptr(), sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), isd.data_ptr(),
                  out.data_ptr(), part.data_ptr(), B, S, S, S, C, 0.01, 1, st)
        _lib.call("tmf_c1_bwd_wgrad_bf16", x.data_ptr(), w.data_ptr(), sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), isd.data_ptr(),
                  coef.data_ptr(), out.data_ptr(), dw.data_ptr(), ws.data_ptr(), nby, B, S, S, S, C, 0.01, 1, 0, st)
        continue
    _lib.call("tmf_c1_stats", x.data_ptr(), w.data_ptr(), part.data_ptr(), B, S, S, S, C, st)
    _lib.call("tmf_c1_bn_pool_fwd", x.data_ptr(), w.data_ptr(), sc.data_ptr(), sh.data_ptr(), out.data_ptr(), B, S, S, S, C, 0.01, st)
    _lib.call("tmf_c1_bwd_reduce", x.data_ptr(), w.data_ptr(), sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), isd.data_ptr(),
              out.data_ptr(), part.data_ptr(), B, S, S, S, C, 0.01, st)
    _lib.call("tmf_c1_bwd_wgrad", x.data_ptr(), w.data_ptr(), sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), isd.data_ptr(),
              coef.data_ptr(), out.data_ptr(), dw.data_ptr(), ws.data_ptr(), nby, B, S, S, S, C, 0.01, 0, st)
torch.cuda.synchronize()
