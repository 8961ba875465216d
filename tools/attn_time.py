#!/usr/bin/env python3
"""Time the attention kernels (forward, backward) at the token counts of the two bench configurations (216 = 96^3 input,
512 = 128^3), B = 8, 4 heads x 32.   python tools/attn_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transmf_ad_amd import _lib          # noqa: E402

dev = "cuda:0"
out = []
for N in (216, 512):
    B, h, dh = 8, 4, 32
    inner = h * dh
    q = torch.randn((B, N, inner), device=dev)
    kv = torch.randn((B, N, 2 * inner), device=dev)
    o = torch.empty_like(q); lse = torch.empty((B, h, N), device=dev)
    do = torch.randn_like(q); dq = torch.empty_like(q); dkv = torch.empty_like(kv)
    st = torch.cuda.current_stream().cuda_stream

    def fwd():
        _lib.call("tmf_xattn_fwd", q.data_ptr(), kv.data_ptr(), kv.data_ptr() + inner * 4, o.data_ptr(), lse.data_ptr(),
                  B, h, N, N, dh, inner, 2 * inner, dh ** -0.5, st)

    def bwd():
        _lib.call("tmf_xattn_bwd", q.data_ptr(), kv.data_ptr(), kv.data_ptr() + inner * 4, o.data_ptr(), lse.data_ptr(),
                  do.data_ptr(), dq.data_ptr(), dkv.data_ptr(), dkv.data_ptr() + inner * 4, B, h, N, N, dh, inner, 2 * inner,
                  2 * inner, dh ** -0.5, st)
    for fn in (fwd, bwd):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record(); torch.cuda.synchronize()
        out.append(f"N={N} {fn.__name__} {e0.elapsed_time(e1) / 50 * 1e3:6.1f} us")
print(" | ".join(out))
