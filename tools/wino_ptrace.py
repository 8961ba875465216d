#!/usr/bin/env python3
"""Phase timeline of the persistent Winograd forward kernel from an instrumented build (tools/build_variant.py ptrace
--flags=-DTMF_WINO_TRACE; TMF_LIB=transmf_ad_amd/libtmf_ptrace.so): shader-clock stamps of the second item of workgroup 77."""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transmf_ad_amd import _lib, ops          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cin", type=int, default=32)
    ap.add_argument("--cout", type=int, default=32)
    ap.add_argument("--S", type=int, default=48)
    ap.add_argument("--B", type=int, default=8)
    a = ap.parse_args()
    dev = "cuda:0"
    x = torch.randn((a.B, a.S, a.S, a.S, a.cin), device=dev)
    w = torch.randn((a.cout, a.cin, 3, 3, 3), device=dev) * (a.cin * 27) ** -0.5
    uf, _ = ops.pack_weights_wino(w, True, False)
    for _ in range(5):
        ops.conv3d_wino_raw(x, uf, a.cin, a.cout, True)
    torch.cuda.synchronize()
    lib = _lib.load()
    fn = lib.tmf_wino_trace_read
    fn.argtypes = [C.c_void_p, C.c_void_p]
    fn.restype = C.c_int
    blocks = np.zeros((8192, 4), dtype=np.int64)
    phases = np.zeros((8, 64), dtype=np.int64)
    assert fn(blocks.ctypes.data, phases.ctypes.data) == 0
    nch = a.cin // 8
    names = {0: "item start", 20: "exchange written", 21: "barrier", 22: "read + d transform", 23: "stores + sums", 24: "statistics / end"}
    for c in range(min(nch, 4)):
        names.update({1 + 4 * c: f"chunk {c} P0 (32 MFMA + rows 0,3)", 2 + 4 * c: f"chunk {c} wait + barrier",
                      3 + 4 * c: f"chunk {c} P1 (32 MFMA + rows 1,2 of next)", 4 + 4 * c: f"chunk {c} wait"})
    print(f"cin {a.cin} cout {a.cout} {a.B}x{a.S}^3 ({nch} chunks; marks of chunks c and c + 4 overwrite each other)")
    for wv in (0, 3):
        t0 = phases[wv, 0]
        print(f"-- wave {wv}")
        prev = t0
        for i in sorted(names):
            t = phases[wv, i]
            if t == 0:
                continue
            print(f"  {names[i]:42s} +{t - prev:6d}   t = {t - t0:6d}")
            prev = t


if __name__ == "__main__":
    main()
