#!/usr/bin/env python3
"""Time the fused token-side Linear launches (tmf_tok_linear_fwd / _bwd_input) at the two bench token counts.
   python tools/tok_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transmf_ad_amd import _lib          # noqa: E402

dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream


def t(fn, reps=100):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for N in (216, 512):
    R = 8 * N
    row = [f"R={R}:"]
    for K, Nout, ln, gelu, res in ((128, 128, True, False, False), (128, 256, False, False, False), (128, 128, False, False, True),
                                   (128, 512, True, True, False), (512, 128, False, False, True)):
        x = torch.randn((R, K), device=dev); w = torch.randn((Nout, K), device=dev) * 0.05; b = torch.randn(Nout, device=dev)
        y = torch.empty((R, Nout), device=dev); r = torch.randn((R, Nout), device=dev)
        g = torch.ones(K, device=dev); be = torch.zeros(K, device=dev); mu = torch.empty(R, device=dev); rs = torch.empty(R, device=dev)
        lo = torch.empty((R, K), device=dev); pre = torch.empty((R, Nout), device=dev)
        us = t(lambda: _lib.call("tmf_tok_linear_fwd", x.data_ptr(), w.data_ptr(), b.data_ptr(), r.data_ptr() if res else None,
                                 y.data_ptr(), R, K, Nout, g.data_ptr() if ln else None, be.data_ptr() if ln else None, 1e-5,
                                 mu.data_ptr() if ln else None, rs.data_ptr() if ln else None, lo.data_ptr() if ln else None,
                                 pre.data_ptr() if gelu else None, st))
        row.append(f"fwd {K}->{Nout}{' ln' if ln else ''}{' gelu' if gelu else ''}{' +res' if res else ''} {us:5.1f} us")
    for Nout, K in ((512, 128), (128, 512), (128, 128), (256, 128)):          # dx[R][K] = dy[R][Nout] . W[Nout][K]
        dy = torch.randn((R, Nout), device=dev); w = torch.randn((Nout, K), device=dev) * 0.05; dx = torch.empty((R, K), device=dev)
        us = t(lambda: _lib.call("tmf_tok_linear_bwd_input", dy.data_ptr(), w.data_ptr(), dx.data_ptr(), R, Nout, K, None, None, None,
                                 None, None, None, None, None, None, 0, st))
        row.append(f"bwd {Nout}->{K} {us:5.1f} us")
    print(" | ".join(row), flush=True)
