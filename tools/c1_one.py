#!/usr/bin/env python3
"""A few launches of ONE first-block pass (for rocprofv3 --pmc): python tools/c1_one.py fwd|bwd [--reps 6] [--split 0|1]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transmf_ad_amd import _lib      # noqa: E402

what = sys.argv[1]
reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 6
split = int(sys.argv[sys.argv.index("--split") + 1]) if "--split" in sys.argv else 1
B, S, C = 8, 96, 32
dev = "cuda:0"
torch.manual_seed(0)
x = torch.rand((B, S, S, S), device=dev)
w = torch.randn((27, C), device=dev) * 0.2
sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
mu, isd = torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) + 0.5
out = torch.empty((B, S // 2, S // 2, S // 2, C), device=dev)
go = torch.randn_like(out)
gb = _lib.query("tmf_c1_gram_bytes", B, S, S, S, C)
gram = torch.empty(gb // 8, device=dev, dtype=torch.float64)
part = torch.empty((max(2, _lib.query("tmf_c1_blocks", B, S, S, S, C)), 2, C), device=dev)
nws = _lib.query("tmf_c1_bwd_fused_workspace_bytes", B, S, S, S, C)
ws = torch.empty(nws // 4, device=dev)
dw, dg, db = torch.empty((C, 27), device=dev), torch.empty(C, device=dev), torch.empty(C, device=dev)
st = torch.cuda.current_stream().cuda_stream
_lib.call("tmf_set_option", b"c1_split", split)
_lib.call("tmf_c1_stats_g", x.data_ptr(), w.data_ptr(), part.data_ptr(), gram.data_ptr(), gb, B, S, S, S, C, st)
for _ in range(reps):
    if what == "fwd":
        _lib.call("tmf_c1_bn_pool_fwd", x.data_ptr(), w.data_ptr(), sc.data_ptr(), sh.data_ptr(), out.data_ptr(), B, S, S, S, C, 0.01, st)
    else:
        _lib.call("tmf_c1_bwd_fused", x.data_ptr(), w.data_ptr(), sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), isd.data_ptr(), go.data_ptr(),
                  gram.data_ptr(), dw.data_ptr(), dg.data_ptr(), db.data_ptr(), ws.data_ptr(), nws, B, S, S, S, C, 0.01, 1, st)
torch.cuda.synchronize()
