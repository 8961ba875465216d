#!/usr/bin/env python3
"""Timeline of the Winograd conv kernel from an instrumented build (-DTMF_WINO_TRACE, loaded with TMF_LIB=...): shader-clock
stamps of every workgroup (start, end, CU) -> duration and the gap to the next workgroup on the same CU; phase stamps of
one workgroup.   TMF_LIB=transmf_ad_amd/libtmf_hip_trace.so python tools/wino_trace.py [--cin 32 --cout 32 --S 48]"""
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
    nb = min(8192, a.B * (a.S // 4) * (a.S // 8) * (a.S // 8))
    bl = blocks[:nb]
    hw, xcc = bl[:, 2], bl[:, 3] & 0xF
    cu = ((hw >> 8) & 0xF) | (((hw >> 12) & 0x1) << 4) | (((hw >> 13) & 0x7) << 5) | (xcc << 8)      # cu_id, sh_id, se_id, xcc
    dur = bl[:, 1] - bl[:, 0]
    print(f"{nb} workgroups on {len(set(cu.tolist()))} CUs; duration cycles: median {np.median(dur):.0f} mean {dur.mean():.0f} "
          f"p10 {np.percentile(dur, 10):.0f} p90 {np.percentile(dur, 90):.0f}")
    gaps = []
    for c in set(cu.tolist()):
        idx = np.where(cu == c)[0]
        o = idx[np.argsort(bl[idx, 0])]
        g = bl[o[1:], 0] - bl[o[:-1], 1]
        gaps.extend(g.tolist())
    gaps = np.array(gaps)
    print(f"gap between consecutive workgroups on a CU (cycles of the shader clock counter): median {np.median(gaps):.0f} mean {gaps.mean():.0f} "
          f"p10 {np.percentile(gaps, 10):.0f} p90 {np.percentile(gaps, 90):.0f}")
    span = bl[:, 1].max() - bl[:, 0].min()
    print(f"whole launch: {span} counter cycles")
    names = {i: f"mark {i}" for i in range(36)}
    names.update({0: "start", 1: "first halo + weights requested", 2: "... in LDS (barrier)", 30: "epilogue start", 31: "exchange written",
                  32: "barrier", 33: "read + d transform", 34: "stores + stat sums", 35: "end"})
    for wv in (0, 3, 4, 7):
        t0 = phases[wv, 0]
        prev = t0
        print(f"-- wave {wv}")
        for i in sorted(names):
            v = phases[wv, i]
            if v == 0:
                continue
            print(f"  {names[i]:34s} +{v - prev:6d}   t = {v - t0:6d}")
            prev = v


if __name__ == "__main__":
    main()
