#!/usr/bin/env python3
"""Per-kernel micro-benchmark on the shapes of one sNet stream (batch B, S^3 input): conv fwd (+stats),
data-gradient, weight-gradient, and the BN/act/pool passes, timed with HIP events on the launch stream.
Usage: python tools/kbench.py [--B 8] [--S 96] [--reps 10] [--waves 4,8] [--out gpurun_out/kbench.json]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transmf_ad_amd import _lib, ops          # noqa: E402

LAYERS = [  # name, cin, cout, k, spatial divisor, pool
    ("conv1.0", 1, 32, 3, 1, "max"), ("conv2.0", 32, 32, 3, 2, None), ("conv2.3", 32, 64, 3, 2, "max"),
    ("conv3.0", 64, 64, 3, 4, None), ("conv3.3", 64, 128, 3, 4, "max"), ("conv4.0", 128, 256, 3, 8, None),
    ("conv4.3", 256, 128, 1, 8, "avg")]


def timeit(fn, reps):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--S", type=int, default=96)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--waves", default="4,8")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    dev = "cuda:0"
    rows = []
    for waves in [int(w) for w in a.waves.split(",")]:
        _lib.call("tmf_set_option", b"conv_waves", waves)
        tot = 0.0
        for name, cin, cout, k, div, pool in LAYERS:
            s = a.S // div
            x = torch.randn((a.B, s, s, s, cin), device=dev)
            w = torch.randn((cout, cin, k, k, k), device=dev) * (cin * k ** 3) ** -0.5
            wp, wd = ops.pack_weight(w), ops.pack_weight_dgrad(w)
            dz = torch.randn((a.B, s, s, s, cout), device=dev)
            flops = 2.0 * cin * cout * k ** 3 * a.B * s ** 3
            r = {"waves": waves, "layer": name, "gflop": flops / 1e9}
            r["fwd_ms"] = timeit(lambda: ops.conv3d_raw(x, wp, cin, cout, k, True), a.reps)
            r["wgrad_ms"] = timeit(lambda: ops.conv3d_wgrad(x, dz, cin, cout, k), a.reps)
            r["dgrad_ms"] = timeit(lambda: ops.conv3d_raw(dz, wd, cout, cin, k, False), a.reps) if cin > 1 else 0.0
            # BN / act / pool passes
            z = dz
            sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
            mu, isd = torch.zeros(cout, device=dev), torch.ones(cout, device=dev)
            pc = _lib.pool_code(pool)
            so = s // 2 if pool else s
            out = torch.empty((a.B, so, so, so, cout), device=dev)
            st = torch.cuda.current_stream().cuda_stream
            r["bn_fwd_ms"] = timeit(lambda: _lib.call("tmf_bn_act_pool_fwd", z.data_ptr(), sc.data_ptr(), sh.data_ptr(),
                                                      out.data_ptr(), a.B, s, s, s, cout, pc, 0.01, st), a.reps)
            nb = _lib.query("tmf_bn_act_pool_bwd_blocks", a.B, s, s, s, cout, pc)
            part = torch.empty((nb, 2, cout), device=dev)
            coef = torch.zeros((2, cout), device=dev)
            dzo = torch.empty_like(z)
            r["bn_red_ms"] = timeit(lambda: _lib.call("tmf_bn_act_pool_bwd_reduce", z.data_ptr(), out.data_ptr(), sc.data_ptr(),
                                                      sh.data_ptr(), mu.data_ptr(), isd.data_ptr(), part.data_ptr(),
                                                      a.B, s, s, s, cout, pc, 0.01, st), a.reps)
            r["bn_app_ms"] = timeit(lambda: _lib.call("tmf_bn_act_pool_bwd_apply", z.data_ptr(), out.data_ptr(), sc.data_ptr(),
                                                      sh.data_ptr(), mu.data_ptr(), isd.data_ptr(), coef.data_ptr(),
                                                      dzo.data_ptr(), a.B, s, s, s, cout, pc, 0.01, st), a.reps)
            if cin == 1:
                xi = x.view(a.B, s, s, s)
                wp1 = wp.view(27, cout)
                nb1 = _lib.query("tmf_c1_blocks", a.B, s, s, s, cout)
                p1 = torch.empty((nb1, 2, cout), device=dev)
                dwo = torch.empty((27, cout), device=dev)
                nby = _lib.query("tmf_c1_bwd_wgrad_workspace_bytes", a.B, s, s, s, cout)
                ws1 = torch.empty((nby // 4,), device=dev)
                f = {}
                f["stats"] = timeit(lambda: _lib.call("tmf_c1_stats", xi.data_ptr(), wp1.data_ptr(), p1.data_ptr(),
                                                      a.B, s, s, s, cout, st), a.reps)
                f["fwd"] = timeit(lambda: _lib.call("tmf_c1_bn_pool_fwd", xi.data_ptr(), wp1.data_ptr(), sc.data_ptr(),
                                                    sh.data_ptr(), out.data_ptr(), a.B, s, s, s, cout, 0.01, st), a.reps)
                f["reduce"] = timeit(lambda: _lib.call("tmf_c1_bwd_reduce", xi.data_ptr(), wp1.data_ptr(), sc.data_ptr(),
                                                       sh.data_ptr(), mu.data_ptr(), isd.data_ptr(), out.data_ptr(),
                                                       p1.data_ptr(), a.B, s, s, s, cout, 0.01, st), a.reps)
                f["wgrad"] = timeit(lambda: _lib.call("tmf_c1_bwd_wgrad", xi.data_ptr(), wp1.data_ptr(), sc.data_ptr(),
                                                      sh.data_ptr(), mu.data_ptr(), isd.data_ptr(), coef.data_ptr(),
                                                      out.data_ptr(), dwo.data_ptr(), ws1.data_ptr(), nby,
                                                      a.B, s, s, s, cout, 0.01, 0, st), a.reps)
                r["fused_c1_ms"] = f
                print(f"   fused conv1 block: stats {f['stats']:.3f} fwd {f['fwd']:.3f} reduce {f['reduce']:.3f} "
                      f"wgrad {f['wgrad']:.3f}  sum {sum(f.values()):.3f} ms  (unfused sum "
                      f"{r['fwd_ms'] + r['wgrad_ms'] + r['bn_fwd_ms'] + r['bn_red_ms'] + r['bn_app_ms']:.3f} ms)", flush=True)
            zb = z.numel() * 4 / 1e9
            r["fwd_tf"] = flops / r["fwd_ms"] / 1e9
            r["wgrad_tf"] = flops / r["wgrad_ms"] / 1e9
            r["dgrad_tf"] = flops / r["dgrad_ms"] / 1e9 if cin > 1 else 0.0
            r["bn_fwd_gbs"] = (zb + out.numel() * 4 / 1e9) / r["bn_fwd_ms"] * 1e3
            r["bn_app_gbs"] = (2 * zb + out.numel() * 4 / 1e9) / r["bn_app_ms"] * 1e3
            rows.append(r)
            t = r["fwd_ms"] + r["wgrad_ms"] + r["dgrad_ms"] + r["bn_fwd_ms"] + r["bn_red_ms"] + r["bn_app_ms"]
            tot += t
            print(f"w{waves} {name:8s} fwd {r['fwd_ms']:7.3f} ms {r['fwd_tf']:6.1f} TF | dgrad {r['dgrad_ms']:7.3f} ms "
                  f"{r['dgrad_tf']:6.1f} TF | wgrad {r['wgrad_ms']:7.3f} ms {r['wgrad_tf']:6.1f} TF | bn fwd {r['bn_fwd_ms']:6.3f} "
                  f"({r['bn_fwd_gbs']:5.0f} GB/s) red {r['bn_red_ms']:6.3f} app {r['bn_app_ms']:6.3f} ({r['bn_app_gbs']:5.0f} GB/s)",
                  flush=True)
            del x, w, dz, out, dzo
        print(f"w{waves} one stream total {tot:.2f} ms  -> both streams {2 * tot:.2f} ms", flush=True)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        json.dump(rows, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
