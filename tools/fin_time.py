#!/usr/bin/env python3
"""Times tmf_bn_finalize / tmf_bn_bwd_finalize on partial-statistic tables of the step's sizes (run once per TMF_LIB)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transmf_ad_amd import _lib
print("library:", os.environ.get("TMF_LIB", "in-tree"))
dev = "cuda:0"
for nblk, C in ((4096, 32), (4096, 64), (1728, 32), (1728, 64), (512, 64), (432, 128), (256, 256)):
    part = torch.randn((nblk, 2, C), device=dev)
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    outs = [torch.empty(C, device=dev) for _ in range(4)]
    dg, db, coef = torch.empty(C, device=dev), torch.empty(C, device=dev), torch.empty((2, C), device=dev)
    st = torch.cuda.current_stream().cuda_stream
    res = []
    for which in (0, 1):
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _i in range(50):
                if which == 0:
                    _lib.call("tmf_bn_finalize", part.data_ptr(), nblk, C, 1e6, g.data_ptr(), b.data_ptr(), None, rm.data_ptr(),
                              rv.data_ptr(), 0.1, 1e-5, outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), outs[3].data_ptr(), st)
                else:
                    _lib.call("tmf_bn_bwd_finalize", part.data_ptr(), nblk, C, 1e6, dg.data_ptr(), db.data_ptr(), coef.data_ptr(), st)
            e1.record(); e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / 50)
        res.append(best * 1e3)
    print(f"nblk {nblk:5d} C {C:4d}: finalize {res[0]:6.1f} us  bwd_finalize {res[1]:6.1f} us (back to back, incl. launch)")
