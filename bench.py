#!/usr/bin/env python3
"""bench.py — volume-pairs/s of the TransMF_AD train step (fwd + bwd + Adam) on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N --steps K --warmup W          (starts the N ranks itself, as child processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = the reference's train_step (kfold_train_adversarial.py:101-136) on one synthetic
batch already resident in HBM: zero_grad, forward of model_ad(dim=128, depth=3, heads=4,
dim_head=32, mlp_dim=512), CE + adversarial CE, backward, (gradient all-reduce over RCCL when
N > 1), Adam step.  Workload (BASELINE.json configs[1]): batch 8 per GPU of 1x96^3 MRI+PET pairs,
fp32.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist
from torch import nn

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_MFMA_TFLOPS = 2500.0     # dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_FP32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 2.4 GHz
PEAK_HBM_GBS = 8000.0


def _config_tag(B, S, precision):
    if (B, S, precision) == (8, 96, "fp32"):
        return " (BASELINE configs[1])"
    if (S, precision) == (128, "bf16"):
        return " (BASELINE configs[2]: 128^3, bf16 MFMA 3D conv)"
    return ""


def conv_flops_per_pair(size, dim=128):
    """Algorithmic conv FLOPs fwd+bwd per MRI+PET pair: 6*sum(F_l) - 2*F_conv1 per
    stream... summed over both streams; no dgrad for conv1.  size: edge length or (D, H, W); floor-mode pooling."""
    q, h, d, d2 = dim // 4, dim // 2, dim, dim * 2
    layers = [(1, q, 3, 1), (q, q, 3, 2), (q, h, 3, 2), (h, h, 3, 4), (h, d, 3, 4), (d, d2, 3, 8), (d2, d, 1, 8)]
    dims = (size, size, size) if isinstance(size, int) else tuple(size)
    tot = 0.0
    for i, (ci, co, k, div) in enumerate(layers):
        vox = 1
        for e in dims:
            for _ in range(div.bit_length() - 1):
                e //= 2
            vox *= e
        f = 2.0 * ci * co * k ** 3 * vox
        tot += f * (2 if i == 0 else 3)
    return 2 * tot


def conv_bytes_per_pair(size, elem, dim=128):
    """Algorithmic HBM bytes fwd+bwd per MRI+PET pair, two-pass-BatchNorm minimum: per conv layer with
    input bytes I, raw output bytes O, post-pool bytes P:  fwd = I + 2O + P;  bwd = 2I + 5O + 2P  (minus I + O on conv1:
    no data gradient); both streams.  elem = 4 (fp32 tensors) or 2 (bf16 storage; SURVEY's 3.094 GB at 128^3 prices the network
    input at 2 B as well although it stays fp32 here — the smaller, conservative figure is used)."""
    q, h, d, d2 = dim // 4, dim // 2, dim, dim * 2
    layers = [(1, q, 0, True), (q, q, 1, False), (q, h, 1, True), (h, h, 2, False), (h, d, 2, True), (d, d2, 3, False),
              (d2, d, 3, True)]
    dims = (size, size, size) if isinstance(size, int) else tuple(size)
    tot = 0.0
    for i, (ci, co, lvl, pooled) in enumerate(layers):
        vox = 1
        pvox = 1
        for e in dims:
            e >>= lvl
            vox *= e
            pvox *= e // 2
        I = vox * ci * elem
        O = vox * co * elem
        P = (pvox if pooled else vox) * co * elem
        tot += (I + 2 * O + P) + (2 * I + 5 * O + 2 * P) - ((I + O) if i == 0 else 0)
    return 2 * tot


def _cd(a, b):
    return -(-a // b)


def wino_exec_flops(kind, B, D, H, W, ci, co):
    """Matrix flops a Winograd launch EXECUTES and the pipe they run on -> (flops, "fp32" | "bf16").  2 * 64 products per 2x2x2
    tile, input and output channel, over the PADDED bricks the library's own plan walks (forward / data gradient:
    tmf_conv3d_wino_bricks x 32 tiles; weight gradient: tmf_conv3d_wgrad_wino_tiles).  Where the split kernel takes the launch
    (csrc/conv3d_winox.hip: every fp32 product as six bf16 partial products) the count is 6 x that, on the bf16 matrix pipe."""
    from transmf_ad_amd import _lib
    if kind == "wgrad":
        return 2.0 * 64 * ci * co * _lib.query("tmf_conv3d_wgrad_wino_tiles", B, D, H, W, ci, co), "fp32"
    base = 2.0 * 64 * ci * co * 32 * _lib.query("tmf_conv3d_wino_bricks2", B, D, H, W, ci, co)
    if _lib.query("tmf_conv3d_wino_kernel_name2", B, D, H, W, ci, co, 0).startswith(b"conv3d_winox"):
        return 6.0 * base, "bf16"
    return base, "fp32"


# the first block's z as exact bf16 splits: K = 27 taps padded to 48, six partial products per fp32 product
C1_SPLIT_FLOP_RATIO = 6.0 * 48.0 / 27.0


def c1_split_on():
    from transmf_ad_amd import _lib
    return bool(_lib.query("tmf_c1_split_mode"))


def exec_flops_per_pair(size, wino, B, dim=128, streams=2, c1_split=False):
    """EXECUTED matrix flops fwd+bwd per pair at batch B, per pipe -> {"fp32": .., "bf16": ..}: the algorithmic count of
    conv_flops_per_pair with the Cin > 1 3x3x3 layers priced by wino_exec_flops where the step runs them in the Winograd form
    (wino: 0 never, 1 data gradients, 2 + forward, 3 + weight gradients) — the numerators of the whole step's `mfma_frac`."""
    q, h, d, d2 = dim // 4, dim // 2, dim, dim * 2
    layers = [(1, q, 3, 0), (q, q, 3, 1), (q, h, 3, 1), (h, h, 3, 2), (h, d, 3, 2), (d, d2, 3, 3), (d2, d, 1, 3)]
    dims = (size, size, size) if isinstance(size, int) else tuple(size)
    tot = {"fp32": 0.0, "bf16": 0.0}
    for i, (ci, co, k, lvl) in enumerate(layers):
        D, H, W = (e >> lvl for e in dims)
        f = 2.0 * ci * co * k ** 3 * D * H * W
        if i == 0 and c1_split:              # csrc/conv1_fused.hip SPLIT: 18 bf16 MFMAs of K = 16 per 32 voxels x 32 channels,
            tot["bf16"] += 2 * f * C1_SPLIT_FLOP_RATIO     # two evaluations of z per step (forward, one-pass backward)
            continue
        if i == 0 or k == 1:
            tot["fp32"] += f * (2 if i == 0 else 3)
            continue
        for kind, a, b, on in (("fwd", ci, co, wino >= 2), ("dgrad", co, ci, wino >= 1), ("wgrad", ci, co, wino >= 3)):
            if on:
                fl, pipe = wino_exec_flops(kind, B, D, H, W, a, b)
                tot[pipe] += fl / B
            else:
                tot["fp32"] += f
    return {k: streams * v for k, v in tot.items()}


def _at_peak_ms(flops_by_pipe):
    """Milliseconds the executed matrix flops take at the peak of the pipe they run on."""
    return flops_by_pipe.get("fp32", 0.0) / PEAK_FP32_MFMA_TFLOPS / 1e9 + flops_by_pipe.get("bf16", 0.0) / PEAK_BF16_MFMA_TFLOPS / 1e9


def _pmc_traffic(kernel, precision, storage, B, vol):
    """HBM bytes per launch of a kernel instance from the committed rocprofv3 --pmc passes (counters cannot be read
    live): profiles/r03_pmc_traffic.json = {"precision|storage|B|DxHxW": {kernel instance: {"hbm_bytes_per_launch": ...}}};
    None for configurations that were not profiled."""
    for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json"):          # newest committed pass first
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                tab = json.load(f).get(f"{precision}|{storage}|{B}|{'x'.join(map(str, vol))}", {})
            for key, row in tab.items():
                if key in kernel:
                    return dict(row, source=f"profiles/{name}")
        except Exception:
            pass
    return None


def _spin_up(dev, seconds):
    """The chip raises its clocks only after a few hundred ms of load (a launch measured straight after start-up is
    ~12 % slower).  Load it with an UNRELATED kernel (a torch matmul) so that neither the live numbers nor a kernel trace
    of this command mix warm-up launches into the conv kernels' statistics."""
    a = torch.randn((4096, 4096), device=dev)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(10):
            torch.mm(a, a)
        torch.cuda.synchronize()


def _time_launches(fn, reps):
    for _ in range(3):             # first calls pay module loading / allocator growth (a --roofline-only run starts cold)
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def measure_conv_launches(ops, _lib, dev, B, vol, precision, storage, reps=10):
    """Every convolution launch of ONE encoder's train step, timed live with HIP events on the launching stream (torch's
    current stream), `reps` back-to-back launches each: forward, data-gradient and weight-gradient (+ its slab
    reduction) of the five Cin > 1 3x3x3 layers, and the four passes of the fused first block as one group.  Returns a
    list of rows {layer, pass, kernel, ms, flops, bytes}: flops / bytes are ALGORITHMIC per launch (2*27*Cin*Cout per
    output voxel; tensors read + written once, at the storage width the step really uses)."""
    dim = 128
    rows = []
    # the Winograd instances as a kernel trace names them (conv3d_wino.hip: the persistent one-wave-per-SIMD kernels and their
    # brick geometry per volume)
    def wino_name(D, H, W, cin_, cout_, stats):
        return _lib.query("tmf_conv3d_wino_kernel_name2", B, D, H, W, cin_, cout_, stats).decode()
    b16 = precision == "bf16"
    s16 = b16 and storage == "bf16"
    adt = torch.bfloat16 if s16 else torch.float32
    el = 2 if s16 else 4
    for name, ci, co, lvl in (("conv2.0", 32, 32, 1), ("conv2.3", 32, 64, 1), ("conv3.0", 64, 64, 2),
                              ("conv3.3", 64, 128, 2), ("conv4.0", 128, 256, 3)):
        D, H, W = (e >> lvl for e in vol)
        vox = B * D * H * W
        x = torch.randn((B, D, H, W, ci), device=dev).to(adt)
        dz = torch.randn((B, D, H, W, co), device=dev).to(adt)
        w = torch.randn((co, ci, 3, 3, 3), device=dev) * (27 * ci) ** -0.5
        fl = 2.0 * 27 * ci * co * vox
        by = float(vox * (ci + co) * el)
        if b16:
            wf, wd = ops.pack_weight_bf16(w), ops.pack_weight_dgrad_bf16(w)
            io = 3 if s16 else 0
            kf = _lib.query("tmf_conv3d_fwd_bf16_kernel_name", B, D, H, W, ci, co, io).decode()
            kd = _lib.query("tmf_conv3d_fwd_bf16_kernel_name", B, D, H, W, co, ci, io).decode()
            fns = (("fwd", kf, lambda: ops.conv3d_bf16_raw(x, wf, ci, co, True, out_bf16=s16)),
                   ("dgrad", kd, lambda: ops.conv3d_bf16_raw(dz, wd, co, ci, False, out_bf16=s16)),
                   ("wgrad", _lib.query("tmf_conv3d_wgrad_bf16_kernel_name", B, D, H, W, ci, co, 1 if s16 else 0).decode(),
                    lambda: ops.conv3d_wgrad_bf16(x, dz, ci, co)))
        else:
            wf, wd = ops.pack_weights_both(w, True)
            kf = _lib.query("tmf_conv3d_fwd_kernel_name", B, D, H, W, ci, co, 3).decode()
            kd = _lib.query("tmf_conv3d_fwd_kernel_name", B, D, H, W, co, ci, 3).decode()
            kw = _lib.query("tmf_conv3d_wgrad_kernel_name", B, D, H, W, ci, co, 3).decode()
            fns = (("fwd", f"conv3d_fwd_kernel<{kf}>", lambda: ops.conv3d_raw(x, wf, ci, co, 3, True)),
                   ("dgrad", f"conv3d_fwd_kernel<{kd}>", lambda: ops.conv3d_raw(dz, wd, co, ci, 3, False)),
                   ("wgrad", f"conv3d_wgrad_kernel<{kw}>", lambda: ops.conv3d_wgrad(x, dz, ci, co, 3)))
            wino = ops.conv_wino_mode() if precision == "fp32" else 0       # as the step runs them (snet_path.hip make_plan)
            if wino >= 2 and ops.wino_ok(ci, co):
                uf, _ = ops.pack_weights_wino(w, True, False)
                fns = (("fwd", wino_name(D, H, W, ci, co, 1), lambda: ops.conv3d_wino_raw(x, uf, ci, co, True)),) + fns[1:]
            if wino >= 1 and ops.wino_ok(co, ci):
                _, ud = ops.pack_weights_wino(w, False, True)
                fns = (fns[0], ("dgrad", wino_name(D, H, W, co, ci, 0), lambda: ops.conv3d_wino_raw(dz, ud, co, ci, False)), fns[2])
            if wino >= 3 and ops.wgrad_wino_ok(ci, co):
                fns = fns[:2] + (("wgrad", _lib.query("tmf_conv3d_wgrad_wino_kernel_name", B, D, H, W, ci, co).decode(),
                                 lambda: ops.conv3d_wgrad_wino(x, dz, ci, co)),)
            if precision == "fp32x":      # forward / data gradient as the step runs them: six bf16 partial products per fp32 product
                w3f = ops.split3_bf16(w.permute(2, 3, 4, 0, 1).contiguous())
                w3d = ops.split3_bf16(w.flip(2, 3, 4).permute(2, 3, 4, 1, 0).contiguous())
                fns = (("fwd", "conv3d_fwd_split_kernel", lambda: ops.conv3d_split_raw(x, w3f, ci, co, True)),
                       ("dgrad", "conv3d_fwd_split_kernel", lambda: ops.conv3d_split_raw(dz, w3d, co, ci, False)), fns[2])
        for pas, kern, fn in fns:
            ms = _time_launches(fn, reps)
            # executed matrix flops: the Winograd form multiplies 64 numbers per 2x2x2 tile, input and output channel (bricks
            # of 4x8x8 voxels, padded); the direct kernels execute the algorithmic count
            ex, pipe = fl, ("bf16" if b16 else "fp32")
            if kern.startswith("conv3d_wino_wgrad"):
                ex, pipe = wino_exec_flops("wgrad", B, D, H, W, ci, co)
            elif kern.startswith("conv3d_wino"):
                ex, pipe = wino_exec_flops(pas, B, D, H, W, *((ci, co) if pas == "fwd" else (co, ci)))
            rows.append(dict(layer=name, **{"pass": pas}, kernel=kern, ms=ms, flops=fl, bytes=by, exec_flops=ex, pipe=pipe))
        del x, dz, w, wf, wd
    # fused first block: statistics + normalise/pool forward, backward reduce + weight gradient (z never stored)
    D, H, W = vol
    q = dim // 4
    x = torch.rand((B, D, H, W, 1), device=dev)
    conv = torch.nn.Conv3d(1, q, 3, padding=1).to(dev)
    bn = torch.nn.BatchNorm3d(q).to(dev)
    go = torch.randn((B, D // 2, H // 2, W // 2, q), device=dev).to(adt)

    def first_block():
        conv.weight.grad = None
        y = ops.conv_bn_act_pool(x, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, True,
                                 pool="max", out_bf16=s16)
        y.backward(go)
    ms = _time_launches(first_block, reps)
    vox = B * D * H * W
    # (fp32: Gram statistics, forward and the one-pass backward = two evaluations of z; "c1_split": on the bf16 pipe, 6 x 48 / 27 of
    # the algorithmic products.  The kernel label keeps its historic spelling: roofline_report excludes the block by it.)
    c1x = precision == "fp32" and c1_split_on()
    rows.append(dict(layer="conv1.0", **{"pass": "fwd+bwd (Gram statistics + 2 recomputing passes)" if not b16 else "fwd+bwd (4 recompute passes)"},
                     kernel="conv1_fused_kernel<0..3>" + (" bf16" if b16 else " split" if c1x else ""), ms=ms, flops=2 * 2.0 * 27 * q * vox,
                     exec_flops=2 * 2.0 * 27 * q * vox * (C1_SPLIT_FLOP_RATIO if c1x else 1.0), pipe="bf16" if (b16 or c1x) else "fp32",
                     bytes=float(2 * vox * 4 + 2 * (vox // 8) * q * el)))
    return rows


def roofline_report(rows, precision, storage, pairs_per_s, world, gf_pair, bytes_pair, B, vol, exec_gf_pair=None):
    """Fold the per-launch rows into the bench line's `roofline` object.  Every `frac` / `achieved` / `mfma_frac` is PHYSICAL:
    the matrix flops the launches EXECUTE against the matrix peak (<= 1 by construction); the SURVEY 8d figure (algorithmic
    flops: 2*27*Cin*Cout per output voxel, which a Winograd launch reaches with 64/216 of the products) sits beside it as
    `algorithmic_*`.
      * main entry  = the time-dominant kernel instance of a step (all of its launches, FLOP-weighted);
      * step_conv   = FLOP-weighted over EVERY conv launch of one step (both encoders);
      * whole_step  = the train step itself (pairs/s x figure per pair) against both ceilings;
      * best_launch = the single best launch;  layers = the full table."""
    peak_tf = PEAK_BF16_MFMA_TFLOPS if precision == "bf16" else PEAK_FP32_MFMA_TFLOPS     # (of the ALGORITHMIC figures' pipe)
    pk = {"fp32": PEAK_FP32_MFMA_TFLOPS, "bf16": PEAK_BF16_MFMA_TFLOPS}

    def peak_ms(r):                 # milliseconds the row's executed flops take at the peak of the pipe they run on
        return r.get("exec_flops", r["flops"]) / pk[r.get("pipe", "fp32")] / 1e9
    groups = {}
    for r in rows:
        g = groups.setdefault(r["kernel"], dict(ms=0.0, flops=0.0, bytes=0.0, exec_flops=0.0, peak_ms=0.0, launches=0, members=[],
                                                pipe=r.get("pipe", "fp32")))
        g["ms"] += r["ms"]; g["flops"] += r["flops"]; g["bytes"] += r["bytes"]; g["launches"] += 1
        g["exec_flops"] += r.get("exec_flops", r["flops"])
        g["peak_ms"] += peak_ms(r)
        g["members"].append(f"{r['layer']} {r['pass']}")
    # the fused first block is four different kernels (+ their finalize launches) timed as one group: it stays in
    # `kernels` / `step_conv`, but the dominant INSTANCE is a single kernel
    dom_name, dom = max(((k, g) for k, g in groups.items() if not k.startswith("conv1_fused_kernel<0..3>")),
                        key=lambda kv: kv[1]["ms"])

    def fracs(g):
        """executed TF (on its pipe), matrix-pipe fraction, algorithmic TF, its fraction, GB/s, its fraction"""
        return (g["exec_flops"] / g["ms"] / 1e9, g["peak_ms"] / g["ms"], g["flops"] / g["ms"] / 1e9, g["flops"] / g["ms"] / 1e9 / peak_tf,
                g["bytes"] / g["ms"] / 1e6, g["bytes"] / g["ms"] / 1e6 / PEAK_HBM_GBS)
    etf, efr, atf, afr, gbs, hfr = fracs(dom)
    bound = "mfma" if dom["peak_ms"] >= dom["bytes"] / PEAK_HBM_GBS / 1e6 else "hbm"     # ms at either ceiling
    tot_ms = sum(r["ms"] for r in rows)
    tot_fl = sum(r["flops"] for r in rows)
    tot_pk = sum(peak_ms(r) for r in rows)
    tot_ex = {pp: sum(r.get("exec_flops", r["flops"]) for r in rows if r.get("pipe", "fp32") == pp) for pp in ("fp32", "bf16")}
    tot_by = sum(r["bytes"] for r in rows)
    best = max((r for r in rows if r["layer"] != "conv1.0"), key=lambda r: peak_ms(r) / r["ms"])
    traffic = _pmc_traffic(dom_name, precision, storage, B, vol)

    def kern_entry(g):
        e = fracs(g)
        return {"launches_per_encoder_step": g["launches"], "ms": round(g["ms"], 4), "avg_launch_ms": round(g["ms"] / g["launches"], 4),
                "pipe": g["pipe"], "tflops": round(e[0], 1), "mfma_frac": round(e[1], 4), "algorithmic_tflops": round(e[2], 1),
                "algorithmic_frac": round(e[3], 4), "hbm_frac": round(e[5], 4)}
    roof = {
        "bound": bound,
        "kernel": dom_name,
        "scope": f"all {dom['launches']} launches of the time-dominant kernel instance in one encoder's train step "
                 f"({', '.join(dom['members'])}); B={B}, volume {'x'.join(map(str, vol))}",
        "achieved": round(etf if bound == "mfma" else gbs, 2),
        "peak": pk[dom["pipe"]] if bound == "mfma" else PEAK_HBM_GBS,
        "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
        "frac": round(efr if bound == "mfma" else hfr, 4),
        "pipe": dom["pipe"],
        "traffic": None if traffic is None else traffic.get("hbm_bytes_per_launch"),
        "traffic_note": None if traffic is None else f"{traffic.get('note')} [{traffic.get('source')}]",
        "launch_ms": round(dom["ms"] / dom["launches"], 4),
        "launch_note": ("a weight-gradient op is the kernel named here + its tmf_slab_reduce launch(es) (+ wino_wgrad_finish_kernel for "
                        "the Winograd form): launch_ms times the op; in a rocprofv3 --stats summary the rows of those kernels add up to it"
                        if "wgrad" in dom_name else "one launch of the kernel named here"),
        "executed_flops_per_launch": dom["exec_flops"] / dom["launches"],
        "algorithmic_flops_per_launch": dom["flops"] / dom["launches"],
        "algorithmic_bytes_per_launch": dom["bytes"] / dom["launches"],
        "mfma_frac": round(efr, 4), "hbm_frac": round(hfr, 4),
        "algorithmic_achieved": round(atf, 2), "algorithmic_frac": round(afr, 4),
        "products_ratio": round(dom["exec_flops"] / dom["flops"], 4),
        "share_of_conv_time": round(dom["ms"] / tot_ms, 3),
        "kernels": {k: kern_entry(g) for k, g in groups.items()},
        "step_conv": {"what": "over every conv launch of one train step (fwd + dgrad + wgrad of the five Cin>1 layers + the fused "
                              "first block; x2 encoders), launches timed back to back; mfma_frac = time of the executed flops at "
                              "the peak of the pipe each launch runs on / measured time",
                      "ms_per_step": round(2 * tot_ms, 3), "tflops_fp32_pipe": round(tot_ex["fp32"] / tot_ms / 1e9, 2),
                      "tflops_bf16_pipe": round(tot_ex["bf16"] / tot_ms / 1e9, 2),
                      "mfma_frac": round(tot_pk / tot_ms, 4),
                      "algorithmic_tflops": round(tot_fl / tot_ms / 1e9, 2),
                      "algorithmic_frac": round(tot_fl / tot_ms / 1e9 / peak_tf, 4),
                      "hbm_frac": round(tot_by / tot_ms / 1e6 / PEAK_HBM_GBS, 4)},
        "best_launch": {"layer": best["layer"], "pass": best["pass"], "kernel": best["kernel"], "pipe": best.get("pipe", "fp32"),
                        "ms": round(best["ms"], 4), "tflops": round(best.get("exec_flops", best["flops"]) / best["ms"] / 1e9, 2),
                        "mfma_frac": round(peak_ms(best) / best["ms"], 4),
                        "algorithmic_frac": round(best["flops"] / best["ms"] / 1e9 / peak_tf, 4)},
        "layers": {f"{r['layer']} {r['pass']}": {"ms": round(r["ms"], 4), "pipe": r.get("pipe", "fp32"),
                                                 "tflops": round(r.get("exec_flops", r["flops"]) / r["ms"] / 1e9, 1),
                                                 "mfma_frac": round(peak_ms(r) / r["ms"], 3),
                                                 "algorithmic_frac": round(r["flops"] / r["ms"] / 1e9 / peak_tf, 3),
                                                 "hbm_frac": round(r["bytes"] / r["ms"] / 1e6 / PEAK_HBM_GBS, 3)}
                   for r in rows},
        "peak_note": ("frac = achieved / peak with achieved = executed_flops_per_launch x launches / (launch_ms x launches): the matrix "
                      "flops the launches EXECUTE on the pipe named in `pipe` (fp32: v_mfma_f32_32x32x2_f32, 157.3 TF; bf16: "
                      "v_mfma_f32_32x32x16_bf16, 2 500 TF).  A direct kernel executes the algorithmic count (SURVEY.md 8d: 2*27*Cin*Cout per "
                      "output voxel); a Winograd F(2x2x2, 3x3x3) launch executes 2*64*Cin*Cout per 2x2x2 tile over its PADDED bricks "
                      "(tmf_conv3d_wino_bricks / tmf_conv3d_wgrad_wino_tiles) — on the fp32 pipe (conv3d_wino_*) or, in the split kernel "
                      "(conv3d_winox_kernel: every fp32 operand as three exact bf16 parts), as SIX bf16 products each on the bf16 pipe; "
                      "`algorithmic_*` is the 8d figure over the same time against the fp32 peak (may exceed 1 for a Winograd launch).  "
                      "Reproduce from a rocprofv3 --kernel-trace --stats summary of `bench.py --roofline-only`: executed_flops_per_launch "
                      "/ AverageNs of the kernel named here / peak"),
    }
    if precision == "fp32x":
        roof["peak_note"] = ("opt-in mode: the forward / data-gradient kernel evaluates every fp32 product as six bf16 partial products on "
                             f"the bf16 matrix cores (its own ceiling: bf16 peak / 6 = {PEAK_BF16_MFMA_TFLOPS / 6:.1f} TF); fractions price "
                             "the fp32 products against the fp32-MFMA peak (157.3 TF) and may exceed 1 for that kernel")
    if pairs_per_s is not None:
        roof["whole_step"] = whole_step_entry(pairs_per_s, world, gf_pair, bytes_pair, precision, exec_gf_pair)
    return roof


def whole_step_entry(pairs_per_s, world, gf_pair, bytes_pair, precision, exec_gf_pair=None):
    """The train step itself against both ceilings: units/s per GPU x the figure per unit — executed matrix flops per pipe
    (mfma_frac = their time at the peaks / the step time, <= 1) and the algorithmic SURVEY.md 8d figure beside it."""
    peak_tf = PEAK_BF16_MFMA_TFLOPS if precision == "bf16" else PEAK_FP32_MFMA_TFLOPS
    per_gpu = pairs_per_s / world
    ex = exec_gf_pair if isinstance(exec_gf_pair, dict) else {("bf16" if precision == "bf16" else "fp32"): (gf_pair if exec_gf_pair is None else exec_gf_pair)}
    pk_s = _at_peak_ms(ex) / 1e3                          # seconds of matrix-pipe time per pair at the peaks
    return {
        "what": "the train step itself: pairs/s per GPU x figure per pair (executed matrix flops per pipe; algorithmic = SURVEY.md 8d)",
        "conv_tflops_fp32_pipe": round(per_gpu * ex.get("fp32", 0.0) / 1e12, 2), "conv_tflops_bf16_pipe": round(per_gpu * ex.get("bf16", 0.0) / 1e12, 2),
        "mfma_frac": round(per_gpu * pk_s, 4),
        "algorithmic_tflops": round(per_gpu * gf_pair / 1e12, 2), "algorithmic_frac": round(per_gpu * gf_pair / 1e12 / peak_tf, 4),
        "hbm_GBps": round(per_gpu * bytes_pair / 1e9, 1), "hbm_frac": round(per_gpu * bytes_pair / 1e9 / PEAK_HBM_GBS, 4),
        "executed_GFLOP_per_pair": {k: round(v / 1e9, 2) for k, v in ex.items()},
        "algorithmic_GFLOP_per_pair": round(gf_pair / 1e9, 2), "algorithmic_GB_per_pair": round(bytes_pair / 1e9, 3),
        "bound": "mfma" if pk_s >= bytes_pair / PEAK_HBM_GBS / 1e9 else "hbm"}


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="pairs per GPU")
    ap.add_argument("--size", type=int, default=96)
    ap.add_argument("--shape", type=int, nargs=3, default=None, metavar=("D", "H", "W"),
                    help="non-cubic volumes, e.g. 91 109 91 (the ADNI volumes of the reference, datasets/ADNI.py:96)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=8, help="pairs in the bounded CPU sample (default 8; 2 above 96^3)")
    ap.add_argument("--cpu-threads", type=int, default=32)
    ap.add_argument("--precision", choices=["fp32", "bf16", "fp32x"], default="fp32",
                    help="bf16: forward/dgrad 3x3x3 convs on the bf16 matrix cores (BASELINE configs[2] mode; not the headline)")
    ap.add_argument("--storage", choices=["fp32", "bf16"], default="fp32",
                    help="bf16 (with --precision bf16): sNet activations between the conv blocks stored as bf16 tensors")
    ap.add_argument("--model", choices=["ad", "cnn", "single"], default="ad",
                    help="ad: model_ad (headline); cnn: model_CNN_ad; single: model_single (MRI only) — BASELINE configs[4]")
    ap.add_argument("--eval", action="store_true",
                    help="time the reference's val_step instead (eval mode, no_grad forward + CE; kfold_train_adversarial.py:144-161)")
    ap.add_argument("--no-fused-adam", action="store_true", help="torch.optim.Adam(fused=False)")
    ap.add_argument("--torch-adam", action="store_true",
                    help="torch.optim.Adam(fused=True) (~10 launches) instead of the one-launch tmf_adam_step")
    ap.add_argument("--from-host", action="store_true",
                    help="feed every step from HOST memory: raw volumes staged in pinned buffers, copied on a side stream "
                         "(double-buffered) and scaled / flipped on the device (transmf_ad_amd.pipeline, the device form of "
                         "datasets/ADNI.py:59-66 + kfold_train_adversarial.py:106-108) — the PCIe-inclusive rate; the default "
                         "(and the headline value) has the batch resident in HBM")
    ap.add_argument("--no-item-sync", action="store_true",
                    help="leave out the reference step's two loss.item() host syncs between forward and backward "
                         "(kfold_train_adversarial.py:127-128); the default step has them")
    ap.add_argument("--roofline-only", action="store_true",
                    help="skip the train-step timing: only the dominant-kernel loop (so that a rocprofv3 --stats run "
                         "of this command averages exactly the launches the roofline entry quotes)")
    ap.add_argument("--dropout", type=float, default=0.0,
                    help="model_ad(dropout=p): Dropout in the fusion block's Transformer instances (options/option.py:39; the "
                         "reference default and the headline are 0)")
    ap.add_argument("--conv-wino", type=int, choices=[0, 1, 2, 3], default=3,
                    help="fp32 3x3x3 convolutions of the encoders: 3 (default, the library's default) forward, data and weight "
                         "gradients in the Winograd form F(2x2x2, 3x3x3) on the fp32 matrix pipe, 2 forward and data gradients, "
                         "1 data gradients only, 0 the direct kernels")
    ap.add_argument("--no-also", action="store_true",
                    help="the default N=1 run measures the other BASELINE configurations in the same process after the headline "
                         "(`also`: configs[2] 128^3 bf16, configs[4] both readings at batch 16, the fp32x mode); this skips them")
    return ap


# The other BASELINE.json configurations, measured in the SAME process after the headline when bench.py runs with its
# default flags on one GPU (so that the driver's one run witnesses them): name -> argument overrides.
ALSO = (
    ("configs[2]: 128^3, batch 8, bf16 MFMA 3D conv + bf16 activation storage",
     dict(precision="bf16", storage="bf16", size=128, batch=8, model="ad", cpu_batch=2, also_kernel_roofline=True)),
    ("configs[4]: model_CNN_ad (dual-modality reading of --model CNN), batch 16, 96^3, fp32",
     dict(model="cnn", batch=16, cpu_batch=2)),
    ("configs[4]: model_single (MRI-only reading), batch 16, 96^3, fp32",
     dict(model="single", batch=16, cpu_batch=2)),
    ("configs[1] workload in the opt-in fp32x mode (fp32-accurate 3-way bf16 split on the bf16 matrix cores)",
     dict(precision="fp32x")),
    ("configs[1] workload with the direct fp32 convolution kernels (--conv-wino 0: rounds 1-3's path)",
     dict(conv_wino=0)),
)


def _is_default_workload(args):
    return (args.model == "ad" and args.precision == "fp32" and args.storage == "fp32" and args.size == 96 and args.batch == 8
            and not args.shape and not args.eval and not args.from_host and not args.roofline_only and not args.no_item_sync
            and not args.no_cpu_baseline and args.steps > 0 and args.dropout == 0.0 and args.conv_wino == 3)


def launch_command(argv, n_gpus, port, python=None):
    """The command `python bench.py --gpus N ...` turns itself into when it was started WITHOUT a torchrun environment:
    one rank per GPU of this node over RCCL, exactly the driver's own multi-GPU launch line (rendezvous on 127.0.0.1)."""
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py")] + list(argv)


def visible_gpu_error(n_gpus, visible):
    """None when `n_gpus` ranks fit on this node, else the message bench.py exits with."""
    if n_gpus <= visible:
        return None
    return (f"bench.py --gpus {n_gpus}: {n_gpus} GPUs requested, {visible} visible on this node "
            f"(one rank per GPU; no oversubscription in the benchmark)")


def self_launch(args, argv):
    """`python bench.py --gpus N` (N > 1) without WORLD_SIZE: start the N ranks as CHILD processes — before this process has
    made any GPU call (torch.cuda.device_count() does not initialise the device on this image; a process that has must never
    exec another program) —, relay their output (rank 0 prints the ONE JSON line) and return the child's exit code."""
    import socket
    import subprocess
    err = visible_gpu_error(args.gpus, torch.cuda.device_count())
    if err:
        raise SystemExit(err)
    with socket.socket() as s:                       # a free rendezvous port
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool (RCCL needs it)
    return subprocess.call(launch_command(argv, args.gpus, port), env=env)


def main():
    import copy
    args = build_parser().parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, sys.argv[1:]))
    from transmf_ad_amd.parallel import init_from_env

    rank, local, world = init_from_env()
    if world != args.gpus:
        if args.gpus != 1 or world != 1:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node must equal --gpus")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback in the product path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if args.storage == "bf16" and args.precision != "bf16":
        raise SystemExit("--storage bf16 needs --precision bf16")

    cpu_cache = {}
    out = run(args, rank, local, world, dev, cpu_cache, brief=False)
    if rank == 0 and world == 1 and not args.no_also and _is_default_workload(args) \
            and os.environ.get("TMF_DDP_FORCE", "0") != "1":
        also = []
        for title, over in ALSO:
            a = copy.copy(args)
            for k, v in over.items():
                setattr(a, k, v)
            try:
                rec = run(a, rank, local, world, dev, cpu_cache, brief=True)
                rec["config_name"] = title
            except Exception as e:                       # an extra record must never cost the headline line
                rec = {"config_name": title, "error": f"{type(e).__name__}: {e}"}
            also.append(rec)
        out["also"] = also
    if world > 1:
        dist.barrier()
    if dist.is_initialized():
        dist.destroy_process_group()
    if rank == 0:
        # the ONE JSON line goes out last: RCCL's version banner sits in the C stdio buffer until then (piped stdout)
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)


def run(args, rank, local, world, dev, cpu_cache, brief=False):
    """One configuration: build the model, time the steps, measure the roofline / CPU baseline legs -> the record (rank 0;
    None elsewhere).  brief: an `also` record — whole-step roofline only (no per-launch loop), no numerics gate."""
    import statistics
    from transmf_ad_amd import model_ad, ops, _lib
    from transmf_ad_amd.parallel import GradAllReduce

    ops.set_conv_precision(args.precision)
    ops.set_activation_storage(args.storage)
    _lib.call("tmf_set_option", b"conv_wino", args.conv_wino)
    torch.manual_seed(0)
    if args.model == "ad":
        net = model_ad(dim=128, depth=3, heads=4, dim_head=32, mlp_dim=512, dropout=args.dropout).to(dev)
    elif args.model == "cnn":
        from transmf_ad_amd import model_CNN_ad
        net = model_CNN_ad(dim=128).to(dev)
    else:
        from transmf_ad_amd import model_single
        net = model_single(128).to(dev)
    if world == 1 and os.environ.get("TMF_DDP_FORCE", "0") == "1":     # overhead measurement of the N>1 machinery
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if not dist.is_initialized():
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        net = GradAllReduce(net)
        if os.environ.get("TMF_DDP_NOSYNC", "0") == "1":       # diagnosis: the process group exists, the wrapper reduces nothing
            net.require_sync = False
    if world > 1:
        net = GradAllReduce(net)
    # same update rule as the reference's getOptimizer (utils/utils.py:38-39: Adam, lr 1e-4, wd 0): every parameter
    # tensor in ONE launch (transmf_ad_amd.optim.Adam -> tmf_adam_step); --torch-adam: torch's multi-tensor form
    if args.torch_adam or args.no_fused_adam:
        opt = torch.optim.Adam(net.parameters(), lr=1e-4, fused=not args.no_fused_adam)
    else:
        from transmf_ad_amd.optim import Adam as OneLaunchAdam
        opt = OneLaunchAdam(net.parameters(), lr=1e-4)
    crit = nn.CrossEntropyLoss()
    # BASELINE.md section 3 "same run also performs the numerics gate": the product's eval-mode forward at the bench's own
    # initial parameters and inputs against the CPU oracle (<= 1e-3 on logits and loss); evaluated in the cpu_baseline leg
    gate_state = None
    want_gate = (rank == 0 and world == 1 and not args.no_cpu_baseline and args.model == "ad" and not args.shape
                 and args.precision == "fp32" and not brief)
    if want_gate:
        gate_state = {k: v.detach().cpu().clone() for k, v in (net.module if hasattr(net, "module") else net).state_dict().items()}
    B, S = args.batch, args.size
    vol = tuple(args.shape) if args.shape else (S, S, S)
    # BASELINE.md section 3 inputs: np.random.RandomState(1234).rand(...) float32 in [0, 1), MRI first, then PET, from one
    # stream (rank r: seed 1234 + r); labels arange(B) % 2.  Generated on the host once, resident in HBM afterwards.
    import numpy as np
    rs_in = np.random.RandomState(1234 + rank)
    mri = mri0 = torch.from_numpy(rs_in.rand(B, 1, *vol).astype(np.float32)).to(dev)
    pet = pet0 = torch.from_numpy(rs_in.rand(B, 1, *vol).astype(np.float32)).to(dev)
    label = label0 = (torch.arange(B, device=dev) % 2).long()
    ones = torch.ones(B, dtype=torch.int64, device=dev)
    zeros = torch.zeros(B, dtype=torch.int64, device=dev)

    feeder = None
    if args.from_host:
        import itertools
        from transmf_ad_amd import DevicePrefetcher
        rs = np.random.RandomState(1234 + rank)
        pool = [dict(MRI=(rs.rand(B, 1, *vol) * 4000.0).astype(np.float32), PET=(rs.rand(B, 1, *vol) * 9.0).astype(np.float32),
                     label=np.arange(B) % 2) for _ in range(3)]              # raw intensities; 3 host batches, cycled
        # the reference's whole train transform (ScaleIntensity, RandFlip 0.3, RandRotate 0.3 / 0.05 rad, RandZoom 0.3 / 0.95-1)
        feeder = iter(DevicePrefetcher(itertools.cycle(pool), device=dev, seed=rank))
        # what the box's host link delivers (pinned -> device, 64 MB copies): the ceiling of this mode
        pin = torch.empty(16 << 20, dtype=torch.float32, pin_memory=True)
        pin.fill_(1.0)                       # touch the pages: an untouched pinned buffer copies at a fantasy rate
        dbuf = torch.empty(16 << 20, dtype=torch.float32, device=dev)
        dbuf.copy_(pin, non_blocking=True)
        torch.cuda.synchronize()
        t_h = time.perf_counter()
        for _ in range(5):
            dbuf.copy_(pin, non_blocking=True)
        torch.cuda.synchronize()
        h2d_gbps = 5 * pin.numel() * 4 / (time.perf_counter() - t_h) / 1e9
        del pin, dbuf

    def val_step():
        net.eval()
        with torch.no_grad():
            if args.model == "single":
                return crit(net(mri), label)
            lo, _dm, _dp = net(mri, pet)
            return crit(lo, label)

    def step():
        if args.eval:
            return val_step()
        net.train()
        if feeder is not None:              # kfold_train_adversarial.py:106-108: this step's batch arrives from the host
            batch = next(feeder)
            mri, pet, label = batch["MRI"], batch["PET"], batch["label"]
        else:
            mri, pet, label = mri0, pet0, label0
        opt.zero_grad()
        if args.model == "single":          # kfold_train_single.py:91-113: plain CE on model_single(MRI)
            loss = crit(net(mri), label)
        else:
            lo, dm, dp = net(mri, pet)
            ce_loss = crit(lo, label)
            ad_loss = (crit(dm, ones) + crit(dp, zeros)) / 2
            if not args.no_item_sync:       # the reference reads both losses on the host BEFORE backward
                ce_loss.item()              # (kfold_train_adversarial.py:127-128: two host syncs per step)
                ad_loss.item()
            loss = ad_loss + ce_loss
        loss.backward()
        opt.step()
        return loss

    mode = "eager"

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    gate_gpu = None
    gate_train = None
    if want_gate:
        net.eval()
        with torch.no_grad():
            lo_g, _dm_g, _dp_g = net(mri0, pet0)
            gate_gpu = (lo_g.cpu(), crit(lo_g, label0).item())
        net.train()
        # train-mode gate (the metric is fwd + bwd): the FIRST train step on the initial parameters — forward, the reference's
        # loss, backward through the very kernels the timed steps run — with fc_cls's two Dropout(0.5) keep-masks fixed to a
        # regenerable pattern (RandomState(99), the pattern of the golden fixtures) so that the CPU leg below can repeat it
        inner = net.module if hasattr(net, "module") else net
        rs_m = np.random.RandomState(99)
        keep = (rs_m.rand(B, 512) >= 0.5, rs_m.rand(B, 64) >= 0.5)

        class _FixedMask(nn.Module):
            def __init__(self, m):
                super().__init__()
                self.m = torch.from_numpy(m).float().to(dev)

            def forward(self, x):
                return x * self.m * 2.0

            def tmf_keep_mask(self, training):      # lets model_ad take its one-launch heads with this fixed mask
                return self.m * 2.0 if training else None
        drops = (inner.fc_cls[3], inner.fc_cls[7])
        inner.fc_cls[3], inner.fc_cls[7] = _FixedMask(keep[0]), _FixedMask(keep[1])
        net.zero_grad(set_to_none=True)
        lo_t, dm_t, dp_t = net(mri0, pet0)
        loss_t = crit(lo_t, label0) + (crit(dm_t, ones) + crit(dp_t, zeros)) / 2
        loss_t.backward()
        gate_train = (lo_t.detach().cpu(), loss_t.item(), inner.mri_cnn.conv2[3].weight.grad.detach().cpu().clone(), keep)
        net.zero_grad(set_to_none=True)
        inner.fc_cls[3], inner.fc_cls[7] = drops
        inner.load_state_dict({k: v.to(dev) for k, v in gate_state.items()})     # BatchNorm buffers back to the initial state
    loss = torch.zeros((), device=dev)
    if args.roofline_only:
        args.warmup, args.steps = 0, 0
    # The first ~30 steps of a process pay one-off host costs that are not part of a step: lazy code-object loading,
    # the caching allocator growing until buffers recycle (hipMalloc synchronises), hipBLASLt heuristics, and for N > 1
    # the RCCL communicator / channel set-up on the first collectives (measured: 11.4 ms/step over steps 6-25 against
    # 7.3 ms from step 30 on in the host-bound bf16 mode; 23 vs 17.8 ms with a 1-rank RCCL group).  They run here as
    # set-up, before the W warm-up steps the contract asks for; the GPU-bound fp32 step does not change with them.
    setup_steps = int(os.environ.get("TMF_BENCH_SETUP_STEPS", "30")) if args.steps > 0 else 0      # (env: profiling runs)
    for _ in range(setup_steps):
        step()
    # (the collector runs HERE, ahead of the W warm-up steps, and stays off until the K timed steps are over: a full collection
    # is tens of milliseconds on a torch heap — inside the timed region it is a step-sized pause, between the fence and the
    # first timed step it idles the GPU long enough for the clocks to drop: the first three timed steps of a process measured
    # 11.3 / 10.1 / 10.0 ms against 9.5 for the rest)
    import gc
    gc.collect()
    gc.disable()
    # one event per step boundary on the stream the steps are issued on (torch's current stream; the side streams of a
    # step join it before the optimizer): per-step times for ms_per_step_min / _median beside the wall-clock mean
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    for _ in range(args.warmup):
        step()
    fence()
    if isinstance(net, GradAllReduce):
        net.timing = True
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        loss = step()
        marks[i + 1].record()
    torch.cuda.synchronize()
    dt_local = max(time.perf_counter() - t0, 1e-9)          # this rank's own time, before the barrier
    gc.enable()
    fence()
    dt = max(time.perf_counter() - t0, 1e-9)
    per_step = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    per_rank = None
    if isinstance(net, GradAllReduce):
        net.timing = False
        ar = net.exposed_allreduce_ms()
        mine = torch.tensor([dt_local / max(args.steps, 1) * 1e3, sum(ar) / max(len(ar), 1), max(ar) if ar else 0.0],
                            device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)] if world > 1 else [mine]
        if world > 1:
            dist.all_gather(allr, mine)
        per_rank = {"ms_per_step": [round(t[0].item(), 3) for t in allr],
                    "allreduce_exposed_ms_mean": [round(t[1].item(), 3) for t in allr],
                    "allreduce_exposed_ms_max": [round(t[2].item(), 3) for t in allr],
                    "collective_bytes": net.last_reduced_bytes, "collective_kinds": net.last_reduced_kinds,
                    "note": "ms_per_step: each rank's own time for the K steps before the final barrier; "
                            "allreduce_exposed: time between the end of backward's compute and the last collective being "
                            "done + scaled (RCCL time that did not hide under backward); collective_*: every all-reduce of "
                            "the last step in launch order — stream / event: a node's flat gradient buffer reduced in place "
                            "behind its stream / behind the encoder's deep-block event, end: in place at the end of backward, "
                            "bucket: packed from .grad at the end of backward"}
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    final_loss = loss.item()
    assert final_loss == final_loss, "loss is NaN"
    if feeder is not None:
        feeder.close()                     # stops and joins the prefetch thread
        feeder = None

    ms_per_step = dt / max(args.steps, 1) * 1e3
    pairs_per_s = world * B * args.steps / dt
    # the timed region is over: give this configuration's memory back before the roofline / CPU legs and the next record
    del net, opt
    torch.cuda.empty_cache()

    # ---- roofline: every conv launch of a step timed live (HIP events on the launching stream), folded into the
    # time-dominant kernel instance (main entry), the FLOP-weighted step figure, and the whole-step fractions ----
    roof = None
    half = 0.5 if args.model == "single" else 1.0
    gf_pair = conv_flops_per_pair(vol) * half
    by_pair = conv_bytes_per_pair(vol, 2 if args.storage == "bf16" else 4) * half
    # executed matrix flops per pair: the Winograd layers at 64 products per tile in the fp32 mode; the bf16 / fp32x modes run the
    # direct form (fp32x: six bf16 partial products per fp32 product, priced as ONE fp32 product each — see peak_note)
    ex_pair = exec_flops_per_pair(vol, args.conv_wino if args.precision == "fp32" else 0, B,
                                  c1_split=args.precision in ("fp32", "fp32x") and c1_split_on())
    ex_pair = {("bf16" if args.precision == "bf16" and k == "fp32" else k): v * half for k, v in ex_pair.items()}
    if rank == 0 and not args.eval:
        if brief and not getattr(args, "also_kernel_roofline", False):
            roof = {"whole_step": whole_step_entry(pairs_per_s, world, gf_pair, by_pair, args.precision, ex_pair)}
            if args.precision == "fp32x":
                roof["peak_note"] = ("fractions price fp32 products against the fp32-MFMA peak (157.3 TF); the forward / "
                                     "data-gradient kernel of this mode runs on the bf16 matrix cores (six partial products)")
        else:
            _spin_up(dev, float(os.environ.get("TMF_ROOF_SPIN_S", "0.5")))
            rows = measure_conv_launches(ops, _lib, dev, B, vol, args.precision, args.storage,
                                         reps=int(os.environ.get("TMF_ROOF_REPS", "10")))
            roof = roofline_report(rows, args.precision, args.storage, pairs_per_s if args.steps > 0 else None, world, gf_pair,
                                   by_pair, B, vol, ex_pair)
            if brief:                       # an `also` record carries the dominant-kernel entry, not the full tables
                for k in ("layers", "kernels", "best_launch"):
                    roof.pop(k, None)

    cpu = None
    gate = None
    vtxt = f"{S}^3" if not args.shape else "x".join(map(str, vol))
    cpu_batch = args.cpu_batch
    if S > 96 and cpu_batch == 8:
        cpu_batch = 2                      # bounded sample (the contract's 10-30 s of CPU work): 128^3 steps are 2.4x as long
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import tmf_oracle as O         # test infrastructure, used ONLY as the timed CPU baseline
        omodel = {"ad": "model_ad", "cnn": "model_CNN_ad", "single": "model_single"}[args.model]
        key = (omodel, cpu_batch, vol)
        if key not in cpu_cache:
            # MKL-DNN conv3d scales poorly past a few dozen threads (256 SMT threads: 85 s/step vs ~8 s on 8): cap at 32
            cpu_cache[key] = O.cpu_train_step_seconds(cpu_batch, vol, steps=3, warmup=1, model=omodel,
                                                      threads=min(args.cpu_threads, os.cpu_count()))
        sec, threads = cpu_cache[key]
        cpu_model = "unknown CPU"
        try:
            with open("/proc/cpuinfo") as f:
                for line in f:
                    if line.startswith("model name"):
                        cpu_model = line.split(":", 1)[1].strip()
                        break
        except OSError:
            pass
        if gate_gpu is not None:
            spec = O.state_spec("model_ad")
            S_or = O.to_state({k: v.numpy() for k, v in gate_state.items()}, spec, requires_grad=False)
            with torch.no_grad():
                lo_c, _d1, _d2 = O.model_ad_forward(S_or, mri0.cpu(), pet0.cpu(), train=False)
                loss_c = torch.nn.functional.cross_entropy(lo_c, label0.cpu()).item()
            dl = (gate_gpu[0] - lo_c).abs().max().item()
            gate = {"what": "eval-mode forward of the product at the bench's initial parameters and inputs vs the CPU oracle "
                            "(BASELINE.md section 3); `train`: the first train step (forward + reference loss + backward) with fixed "
                            "fc_cls keep-masks vs the oracle's train step on the same inputs, parameters and masks",
                    "max_abs_dlogits": dl, "abs_dloss": abs(gate_gpu[1] - loss_c), "gpu_loss": gate_gpu[1], "cpu_loss": loss_c,
                    "tolerance": 1e-3, "pass": bool(dl <= 1e-3 and abs(gate_gpu[1] - loss_c) <= 1e-3)}
            if gate_train is not None:
                lo_t, loss_t, gw_t, keep = gate_train
                S_tr = O.to_state({k: v.numpy() for k, v in gate_state.items()}, spec, requires_grad=True)
                lo_c, dm_c, dp_c = O.model_ad_forward(S_tr, mri0.cpu(), pet0.cpu(), train=True,
                                                      dropout_masks=(torch.from_numpy(keep[0]), torch.from_numpy(keep[1])))
                loss_tc = O.adversarial_loss(lo_c, dm_c, dp_c, label0.cpu())
                loss_tc.backward()
                gw_c = S_tr["mri_cnn.conv2.3.weight"].grad
                dlt = (lo_t - lo_c.detach()).abs().max().item()
                gerr = ((gw_t - gw_c).abs().max() / gw_c.abs().max()).item()
                gate["train"] = {"max_abs_dlogits": dlt, "abs_dloss": abs(loss_t - loss_tc.item()), "gpu_loss": loss_t,
                                 "cpu_loss": loss_tc.item(), "conv2.3_weight_grad_rel_to_max": gerr,
                                 "tolerance": {"logits": 1e-3, "loss": 1e-3, "weight_grad_rel_to_max": 2e-2},
                                 "pass": bool(dlt <= 1e-3 and abs(loss_t - loss_tc.item()) <= 1e-3 and gerr <= 2e-2)}
                gate["pass"] = bool(gate["pass"] and gate["train"]["pass"])
        unit_txt = "volumes" if args.model == "single" else "pairs"
        cpu = {"value": round(cpu_batch / sec, 4), "unit": "volumes/s" if args.model == "single" else "volume-pairs/s",
               "cores": threads, "kind": "port",
               "sample": f"oracle {omodel} train-mode fwd+bwd (exact fp32 on the host whatever the GPU mode), batch {cpu_batch} of "
                         f"1x{vtxt} {unit_txt} (1/{max(1, B // cpu_batch)} of one batch-{B} step), 1 warm-up + 3 timed steps, "
                         f"{sec:.2f} s/step, {cpu_model}, host cpu_count={os.cpu_count()}, torch threads={threads}"}

    if rank != 0:
        return None
    gf = gf_pair
    if args.eval:       # forward only: sum of F_l per stream
        gf = gf * (32.219 / 95.125) if vol == (96, 96, 96) else gf / 3.0
    model_desc = {"ad": "model_ad(dim=128,depth=3,heads=4,dim_head=32,mlp_dim=512" + (f",dropout={args.dropout}" if args.dropout else "") + ")",
                  "cnn": "model_CNN_ad(dim=128) [BASELINE configs[4], dual-modality reading of --model CNN]",
                  "single": "model_single(128), MRI only [BASELINE configs[4], single-modality reading]"}[args.model]
    out = {
        "metric": ((f"volume-pairs/sec fwd+bwd(+Adam), {vtxt} MRI+PET batch={B} per GPU" if args.model != "single"
                    else f"volumes/sec fwd+bwd(+Adam), {vtxt} MRI only batch={B} per GPU") if not args.eval else
                   f"volume-pairs/sec eval forward (val_step), {vtxt} batch={B} per GPU"),
        "value": round(pairs_per_s, 3), "unit": "volume-pairs/s" if args.model != "single" else "volumes/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "ms_per_step_min": round(min(per_step), 3) if per_step else None,
        "ms_per_step_median": round(statistics.median(per_step), 3) if per_step else None,
        "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        "dtype": {"fp32": "f32", "bf16": ("bf16 MFMA operands (conv fwd/dgrad/wgrad), f32 accumulate, bf16 activation storage between conv blocks"
                           if args.storage == "bf16" else "bf16 MFMA operands (conv fwd/dgrad/wgrad), f32 accumulate+storage"),
                  "fp32x": "f32 via exact 3-way bf16 split on the bf16 MFMA (conv fwd/dgrad; wgrad exact f32 MFMA)"}[args.precision],
        "data": "synthetic",
        "config": {"workload": f"{model_desc} {'val_step' if args.eval else 'train step'}, "
                               f"batch {B} {'volumes' if args.model == 'single' else 'pairs'} of 1x"
                               + (f"{S}^3" if not args.shape else "x".join(map(str, vol))) + f" per GPU, {args.precision}"
                               + (_config_tag(B, S, args.precision) if args.model == "ad" else ""),
                   "global_batch": B * world, "parallelism": f"dp{world}",
                   "step": "val_step: eval-mode no_grad forward + CE" if args.eval else
                           ("zero_grad+fwd+loss+bwd+allreduce+Adam" if args.no_item_sync or args.model == "single" else
                            "zero_grad+fwd+loss+2x loss.item()+bwd+allreduce+Adam (kfold_train_adversarial.py:101-136)"),
                   "conv_algorithm": ({3: "fp32: Winograd F(2x2x2,3x3x3) for forward, data and weight gradients of the Cin>1 3x3x3 blocks "
                                          "(64/216 of the products): weight gradients and the 12^3 block on the fp32 matrix pipe, "
                                          + ("forward / data gradients of the blocks with Cin % 32 == 0 on the bf16 matrix pipe with every fp32 "
                                             "operand split EXACTLY into three bf16 numbers (six partial products, fp32 accumulation; the "
                                             "dropped three are below 2^-24 of the product) — csrc/conv3d_winox.hip; "
                                             if _lib.query("tmf_wino_x_mode") else "forward / data gradients on the fp32 matrix pipe; ")
                                          + ("the first block's z from exact 3-way bf16 splits of the volume and the taps on the bf16 pipe "
                                             "(csrc/conv1_fused.hip SPLIT), " if c1_split_on() else "direct implicit GEMM for the first block, ")
                                          + "direct implicit GEMM for the 1x1x1 block",
                                       2: "fp32: Winograd F(2x2x2,3x3x3) on the fp32 matrix pipe for forward and data gradients of the "
                                          "Cin>1 3x3x3 blocks (exact-fp32 arithmetic, 64/216 of the products), direct implicit GEMM for "
                                          "the weight gradients, the first block and the 1x1x1 block",
                                       1: "fp32: Winograd F(2x2x2,3x3x3) data gradients, direct implicit GEMM otherwise",
                                       0: "fp32: direct implicit GEMM (fp32 MFMA) everywhere"}[args.conv_wino]
                                      if args.precision == "fp32" and not args.eval else "direct implicit GEMM"),
                   "dispatch": mode,
                   "optimizer": ("transmf_ad_amd.optim.Adam (one launch: tmf_adam_step)" if not (args.torch_adam or args.no_fused_adam)
                                 else "torch.optim.adam.Adam"),
                   "input": ("host: raw volumes -> H2D on a copy stream (double-buffered) -> device ScaleIntensity + RandFlip(0.3) "
                             f"+ RandRotate(0.3, 0.05 rad) + RandZoom(0.3, 0.95-1), every step (PCIe-inclusive); {2 * B * vol[0] * vol[1] * vol[2] * 4 / 1e6:.1f} "
                             f"MB per step over a host link measured at {h2d_gbps:.1f} GB/s pinned -> device on this box"
                             if args.from_host else "resident in HBM"),
                   "setup_steps_untimed": setup_steps,
                   "per_step_timing": "ms_per_step = wall clock of the K steps / K (the contract); _min / _median from one HIP "
                                      "event per step boundary on the issuing stream"},
        "conv_tflops_whole_step": round(pairs_per_s / world * (sum(ex_pair.values()) if not args.eval else gf) / 1e12, 2),
        "algorithmic_conv_tflops_whole_step": round(pairs_per_s / world * gf / 1e12, 2),
        "loss": round(final_loss, 6),
        "roofline": roof, "cpu_baseline": cpu,
    }
    if cpu is not None and gate is not None:
        out["numerics_gate"] = gate
    if per_rank is not None:
        out["per_rank"] = per_rank
    if os.environ.get("TMF_BENCH_STEP_TIMES", "0") == "1":      # diagnosis: every timed step's own time
        out["ms_per_step_list"] = [round(t, 3) for t in per_step]
    if brief:           # an `also` record: the keys the headline carries, minus the contract boilerplate
        for k in ("n_gpus", "higher_is_better", "scaling", "vs_baseline", "data", "conv_tflops_whole_step"):
            out.pop(k, None)
        out["config"] = {"workload": out["config"]["workload"], "step": out["config"]["step"],
                         "conv_algorithm": out["config"]["conv_algorithm"], "setup_steps_untimed": setup_steps}
    return out


if __name__ == "__main__":
    main()
