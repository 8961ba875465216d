#!/usr/bin/env python3
"""Time (only) of the Winograd forward / data-gradient / weight-gradient kernels on the encoder's layer shapes — for A/B runs of
kernel variants built with tools/build_variant.py and loaded through TMF_LIB (ablation builds give wrong results on purpose; an
optional --check compares z with the fp64 reference).   TMF_LIB=... python tools/wino_time.py [--what fwd,dgrad,wgrad] [--check]"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transmf_ad_amd import ops          # noqa: E402

SHAPES = [("conv2.0", 32, 32, 48), ("conv2.3", 32, 64, 48), ("conv3.0", 64, 64, 24), ("conv3.3", 64, 128, 24), ("conv4.0", 128, 256, 12)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--what", default="fwd")
    ap.add_argument("--only", default="")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--tag", default=os.path.basename(os.environ.get("TMF_LIB", "libtmf_hip.so")))
    a = ap.parse_args()
    dev = "cuda:0"
    torch.manual_seed(0)
    out = []
    for name, cin, cout, s in SHAPES:
        if a.only and a.only not in name:
            continue
        for what in a.what.split(","):
            ci, co = (cout, cin) if what == "dgrad" else (cin, cout)
            x = torch.randn((a.B, s, s, s, ci), device=dev)
            if what == "wgrad":
                dz = torch.randn((a.B, s, s, s, co), device=dev)
                fn = lambda: ops.conv3d_wgrad_wino(x, dz, ci, co)
            else:
                wt = torch.randn((co, ci, 3, 3, 3), device=dev) * (ci * 27) ** -0.5
                uf, _ = ops.pack_weights_wino(wt, True, False)
                fn = lambda: ops.conv3d_wino_raw(x, uf, ci, co, what == "fwd")
            err = ""
            if a.check:
                if what == "wgrad":
                    dw = fn()
                    r = torch.nn.grad.conv3d_weight(x[:2].double().permute(0, 4, 1, 2, 3), (co, ci, 3, 3, 3), dz[:2].double().permute(0, 4, 1, 2, 3), padding=1)
                    d2 = ops.conv3d_wgrad_wino(x[:2].contiguous(), dz[:2].contiguous(), ci, co)
                    r = r.permute(2, 3, 4, 1, 0).reshape(27, ci, co)
                    err = f" err {(d2.double() - r).abs().max().item() / r.abs().max().item():.1e}"
                else:
                    z = ops.conv3d_wino_raw(x[:1].contiguous(), uf, ci, co, what == "fwd")[0]
                    r = F.conv3d(x[:1].double().permute(0, 4, 1, 2, 3), wt.double(), None, 1, 1).permute(0, 2, 3, 4, 1)
                    err = f" err {(z.double() - r).abs().max().item() / r.abs().max().item():.1e}"
            best = 1e9
            for _ in range(a.rounds):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for i in range(a.reps + 2):
                    if i == 2:
                        e0.record()
                    fn()
                e1.record()
                e1.synchronize()
                best = min(best, e0.elapsed_time(e1) / a.reps)
            exe = 2.0 * a.B * ((s + 1) // 2) ** 3 * 64 * ci * co             # executed products x 2 (unpadded tiles)
            out.append(f"{name} {what} {best * 1e3:6.1f}us({exe / best / 1e9 / 157.3:.2f}){err}")
    print(f"{a.tag:22s} " + "  ".join(out), flush=True)


if __name__ == "__main__":
    main()
