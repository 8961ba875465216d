#!/usr/bin/env python3
"""Phase timeline of the split Winograd kernel from an instrumented build (tools/build_variant.py xtrace --replace
conv3d_winox.hip=transmf_ad_amd/csrc/conv3d_winox.hip --flags=-DTMF_WINOX_TRACE; TMF_LIB=transmf_ad_amd/libtmf_xtrace.so):
shader-clock stamps of the second item of workgroup 77, all eight waves (w and w + 4 share a SIMD)."""
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
    fn = _lib.load().tmf_winox_trace_read
    fn.argtypes = [C.c_void_p]
    fn.restype = C.c_int
    ph = np.zeros((8, 64), dtype=np.int64)
    assert fn(ph.ctypes.data) == 0
    names, order = {}, []
    for c in range(2):
        b = 20 * c
        names.update({b: f"chunk {c}: start", b + 1: f"chunk {c}: copies issued + rows of group a", b + 2: f"chunk {c}: positions 0-3", b + 3: f"chunk {c}: rows of group b",
                      b + 4: f"chunk {c}: positions 4-7", b + 5: f"chunk {c}: end"})
        order += [b, b + 1, b + 2, b + 3, b + 4, b + 5]
    order += [40, 41, 42, 43, 44, 45, 46, 47]
    names.update({40: "exchange written (ho 0)", 41: "barrier", 42: "read + d + stores (ho 0)", 43: "barrier", 44: "exchange written (ho 1)",
                  45: "barrier", 46: "read + d + stores (ho 1)", 47: "barrier"})
    t0 = ph[:, 0].min()
    print(f"cin {a.cin} cout {a.cout} {a.B}x{a.S}^3; cycles since the first wave's chunk start; columns = waves 0..7 (w, w + 4 on one SIMD)")
    for i in order:
        if not ph[:, i].any():
            continue
        print(f"{names[i]:34s} " + " ".join(f"{int(ph[wv, i] - t0):7d}" for wv in range(8)))


if __name__ == "__main__":
    main()
