#!/usr/bin/env python3
"""Phase timeline of the fused fusion kernels (csrc/xformer_fused.hip): per-wave shader-clock stamps of the LAST
instance's forward / backward launches + event timing of whole fusion passes.   python tools/xf_trace.py [N] [B]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transmf_ad_amd import _lib, networks, ops      # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 216
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = "cuda:0"
torch.manual_seed(0)
fz = networks.CrossTransformer_MOD_AVG(128, 3, 4, 32, 512, 0.).to(dev).train()
m0 = torch.randn(B, N, 128, device=dev)
p0 = torch.randn(B, N, 128, device=dev)
go = torch.randn(B, 512, device=dev)


def one(trace=False):
    fz.zero_grad()
    m, p = m0.clone().requires_grad_(True), p0.clone().requires_grad_(True)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    c = fz(m, p)
    e[1].record()
    c.backward(go)
    e[2].record()
    torch.cuda.synchronize()
    return e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])


for fused in (True, False):
    ops.FUSION_FUSED_KERNELS = fused
    for _ in range(5):
        one()
    ts = np.array([one() for _ in range(20)])
    print(f"N={N} B={B} fused={fused}: fwd {np.median(ts[:, 0]) * 1e3:.0f} us  bwd {np.median(ts[:, 1]) * 1e3:.0f} us (median of 20, events)")
ops.FUSION_FUSED_KERNELS = True
tiles = (N + 15) // 16
bufs = [torch.zeros(B * tiles * 4 * 16, dtype=torch.int64, device=dev) for _ in range(3)]
_lib.load().tmf_debug_xf_trace(bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr())
one()
_lib.load().tmf_debug_xf_trace(None, None, None)
names = [["P0 load+LN1", "P1 q", "P2 S=KQ", "softmax", "PV+store", "P4 out-proj", "LN2", "P5 FF1+GELU", "P6 FF2", "LNf", "P7 kv_next"],
         ["S1 LNf'", "S2 dg,dh", "S3 df", "S4 LN2'", "S5 dout", "S6 attn dq", "S7 da", "S8 LN1'"],
         ["attn dk,dv", "dctx"]]
for k, (buf, nm) in enumerate(zip(bufs, names)):
    t = buf.cpu().numpy().reshape(B * tiles, 4, 16).astype(np.int64)
    n = len(nm)
    d = np.diff(t[:, :, :n + 1], axis=2)              # [wg][wave][phase]
    tot = t[:, :, n] - t[:, :, 0]
    print(f"--- {['xf_fwd', 'xf_bwd_q', 'xf_bwd_kv'][k]}: per-wave cycles, median over workgroups (max over waves) | full tiles only")
    full = np.arange(B * tiles) % tiles != tiles - 1 if N % 16 else np.ones(B * tiles, bool)
    for i, name in enumerate(nm):
        print(f"  {name:14s} {np.median(d[full, :, i].max(axis=1)):9.0f}   min {d[full, :, i].min():7d}  max {d[full, :, i].max():7d}")
    print(f"  {'total':14s} {np.median(tot[full].max(axis=1)):9.0f}   launch span (first start -> last end) {t[:, :, n].max() - t[:, :, 0].min()}")
