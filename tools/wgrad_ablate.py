#!/usr/bin/env python3
"""Timing ablations of the transposing-read bf16 weight-gradient kernel (results are garbage): which part of a launch the
waves spend waiting on.  The switches are compile-time (-DTMF_ABLATE_WG=bits; a run-time flag changes the register
allocation of the whole kernel), so this tool first builds one library per variant HERE (no GPU needed):
    python tools/wgrad_ablate.py --build            # -> tools/_alt/libtmf_wg<bits>.so
and then times them on the GPU box, one process per library (TMF_LIB):
    python tools/wgrad_ablate.py [--ci 32 --co 64 --s 64]"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ALT = os.path.join(ROOT, "tools", "_alt")
VARIANTS = ((0, "full kernel"), (1, "no staging copies"), (2, "no LDS fragment reads"), (3, "no copies, no fragment reads"),
            (4, "no MFMAs (fragment reads kept)"), (6, "copies only (no reads, no MFMAs)"), (8, "no barriers"), (11, "MFMAs only"), (15, "skeleton"))

ap = argparse.ArgumentParser()
ap.add_argument("--build", action="store_true")
ap.add_argument("--one", type=int, default=None, help="(internal) time the library in TMF_LIB")
ap.add_argument("--ci", type=int, default=32)
ap.add_argument("--co", type=int, default=64)
ap.add_argument("--s", type=int, default=64)
ap.add_argument("--B", type=int, default=8)
a = ap.parse_args()

if a.build:
    from transmf_ad_amd import build as B
    B.build(verbose=False)
    os.makedirs(ALT, exist_ok=True)
    hipcc = B._hipcc()
    src = os.path.join(B.CSRC, "conv3d_bf16.hip")
    others = [os.path.join(B.CSRC, s.replace(".hip", ".o")) for s in B.SOURCES if s != "conv3d_bf16.hip"]
    procs = []
    for bits, _ in VARIANTS:
        o = os.path.join(ALT, f"conv3d_bf16_wg{bits}.o")
        procs.append((bits, o, subprocess.Popen([hipcc, "-x", "hip", "-c", src, "-o", o, f"-DTMF_ABLATE_WG={bits}"] + B.FLAGS)))
    for bits, o, p in procs:
        if p.wait() != 0:
            sys.exit(f"hipcc failed on variant {bits}")
        subprocess.check_call([hipcc, "-shared", "-o", os.path.join(ALT, f"libtmf_wg{bits}.so"), o] + others + ["--offload-arch=gfx950"])
        os.remove(o)
    print("built", [f"libtmf_wg{b}.so" for b, _ in VARIANTS])
    sys.exit(0)

if a.one is None:
    for bits, name in VARIANTS:
        lib = os.path.join(ALT, f"libtmf_wg{bits}.so")
        if not os.path.exists(lib):
            sys.exit(f"{lib} missing: run  python tools/wgrad_ablate.py --build  first")
        env = dict(os.environ, TMF_LIB=lib)
        out = subprocess.run([sys.executable, __file__, "--one", str(bits), "--ci", str(a.ci), "--co", str(a.co), "--s", str(a.s),
                              "--B", str(a.B)], env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("us=")]
        if not line:
            sys.exit(out.stdout + out.stderr)
        us = float(line[0][3:])
        fl = 2.0 * 27 * a.ci * a.co * a.B * a.s ** 3
        print(f"{name:34s} {us:8.1f} us   {fl / us / 1e6:7.0f} TF-equivalent", flush=True)
    sys.exit(0)

import torch                                  # noqa: E402
from transmf_ad_amd import ops                # noqa: E402
dev = "cuda:0"
x = torch.randn((a.B, a.s, a.s, a.s, a.ci), device=dev).bfloat16()
dz = torch.randn((a.B, a.s, a.s, a.s, a.co), device=dev).bfloat16()
aa = torch.randn((4096, 4096), device=dev)
for _ in range(200):
    torch.mm(aa, aa)
for _ in range(3):
    ops.conv3d_wgrad_bf16(x, dz, a.ci, a.co)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.conv3d_wgrad_bf16(x, dz, a.ci, a.co)
e1.record(); torch.cuda.synchronize()
print(f"us={e0.elapsed_time(e1) / 20 * 1e3}")
