#!/usr/bin/env python3
"""Times of the first block's passes as the train step runs them (fp32 mode): tmf_c1_stats_g (Gram kernels), tmf_c1_bn_pool_fwd and
tmf_c1_bwd_fused, with z as exact bf16 splits ("c1_split" 1, the default) and on the fp32 matrix instructions ("c1_split" 0).
python tools/c1_time.py [--B 8 --S 96 --reps 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transmf_ad_amd import _lib      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--S", type=int, default=96)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    a = ap.parse_args()
    B, S, C = a.B, a.S, 32
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
    calls = {
        "stats_g": lambda: _lib.call("tmf_c1_stats_g", x.data_ptr(), w.data_ptr(), part.data_ptr(), gram.data_ptr(), gb, B, S, S, S, C, st),
        "fwd": lambda: _lib.call("tmf_c1_bn_pool_fwd", x.data_ptr(), w.data_ptr(), sc.data_ptr(), sh.data_ptr(), out.data_ptr(), B, S, S, S,
                                 C, 0.01, st),
        "bwd_fused": lambda: _lib.call("tmf_c1_bwd_fused", x.data_ptr(), w.data_ptr(), sc.data_ptr(), sh.data_ptr(), mu.data_ptr(),
                                       isd.data_ptr(), go.data_ptr(), gram.data_ptr(), dw.data_ptr(), dg.data_ptr(), db.data_ptr(),
                                       ws.data_ptr(), nws, B, S, S, S, C, 0.01, 1, st),
    }
    calls["stats_g"]()
    best = {}
    for _ in range(a.rounds):
        for split in (1, 0):
            _lib.call("tmf_set_option", b"c1_split", split)
            for name, f in calls.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for i in range(a.reps + 2):
                    if i == 2:
                        e0.record()
                    f()
                e1.record()
                e1.synchronize()
                k = (name, split)
                best[k] = min(best.get(k, 1e9), e0.elapsed_time(e1) / a.reps * 1e3)
    _lib.call("tmf_set_option", b"c1_split", 1)
    for name in calls:
        print(f"{name:10s} split {best[(name, 1)]:7.1f} us   fp32 {best[(name, 0)]:7.1f} us   x{best[(name, 0)] / best[(name, 1)]:.2f}", flush=True)


if __name__ == "__main__":
    main()
