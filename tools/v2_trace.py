#!/usr/bin/env python3
"""Phase timeline of ONE workgroup of the LDS-DMA bf16 forward kernel (a -DTMF_TRACE=<workgroup> build of the library,
loaded through TMF_LIB): shader-clock stamps of wave 0 (weight copier) and wave 4 (halo copier).

    python tools/v2_trace.py conv2.0 [--size 128]
"""
import argparse, ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transmf_ad_amd import _lib

NAMES = {1: "entry", 2: "descriptors done", 3: "first weight copies issued", 10: "chunk: halo buffer free", 11: "halo copies issued",
         12: "halo landed", 20: "stage top", 21: "weights landed (own share)", 22: "barrier passed", 23: "next stage issued",
         30: "products issued -> epilogue", 31: "stores issued", 40: "end"}
LAYERS = {"conv2.0": (32, 32, 2), "conv2.3d": (64, 32, 2), "conv2.3": (32, 64, 2), "conv3.3": (64, 128, 4)}
ap = argparse.ArgumentParser()
ap.add_argument("layer", choices=sorted(LAYERS))
ap.add_argument("--size", type=int, default=128)
a = ap.parse_args()
cin, cout, div = LAYERS[a.layer]
B, S = 8, a.size // div
dev = "cuda:0"
x = torch.randn((B, S, S, S, cin), device=dev).bfloat16()
w = (torch.randn((27, cout, cin), device=dev) * 0.05).bfloat16()
z = torch.empty((B, S, S, S, cout), device=dev, dtype=torch.bfloat16)
nb = _lib.query("tmf_conv3d_bf16_stat_blocks", B, S, S, S)
part = torch.empty((nb, 2, cout), device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(5):
    _lib.call("tmf_conv3d_fwd_bf16_t", x.data_ptr(), w.data_ptr(), z.data_ptr(), part.data_ptr(), B, S, S, S, cin, cout, 3, st)
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros((2, 256), dtype=np.uint64)
lib.tmf_debug_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert lib.tmf_debug_trace_read(buf.ctypes.data, buf.nbytes) == 0
for slot, who in ((0, "wave 0 (weights)"), (1, "wave 4 (halo)")):
    ev = [(int(v) >> 48, int(v) & 0xFFFFFFFFFFFF) for v in buf[slot] if int(v) >> 48]
    # the buffer keeps the last launch; entries beyond 'end' are leftovers
    out = []
    for i, (k, t) in enumerate(ev):
        out.append((k, t))
        if k == 40:
            break
    t0 = out[0][1]
    print(f"--- {who}: {a.layer} {cin}->{cout} @{S}^3, {len(out)} events, total {out[-1][1] - t0} cycles")
    prev = t0
    for k, t in out:
        print(f"  +{t - prev:7d}  @{t - t0:7d}  {NAMES.get(k, k)}")
        prev = t
