#!/usr/bin/env python3
"""bench.py — volume-pairs/s of the TransMF_AD train step (fwd + bwd + Adam) on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
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


def _profiled_traffic():
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (counters cannot be read live)."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic_conv2.3_fwd.json")) as f:
            return float(json.load(f)["hbm_bytes_per_launch"])
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="pairs per GPU")
    ap.add_argument("--size", type=int, default=96)
    ap.add_argument("--shape", type=int, nargs=3, default=None, metavar=("D", "H", "W"),
                    help="non-cubic volumes, e.g. 91 109 91 (the ADNI volumes of the reference, datasets/ADNI.py:96)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=8, help="pairs in the bounded CPU sample")
    ap.add_argument("--cpu-threads", type=int, default=32)
    ap.add_argument("--precision", choices=["fp32", "bf16", "fp32x"], default="fp32",
                    help="bf16: forward/dgrad 3x3x3 convs on the bf16 matrix cores (BASELINE configs[2] mode; not the headline)")
    ap.add_argument("--storage", choices=["fp32", "bf16"], default="fp32",
                    help="bf16 (with --precision bf16): sNet activations between the conv blocks stored as bf16 tensors")
    ap.add_argument("--model", choices=["ad", "cnn", "single"], default="ad",
                    help="ad: model_ad (headline); cnn: model_CNN_ad; single: model_single (MRI only) — BASELINE configs[4]")
    ap.add_argument("--eval", action="store_true",
                    help="time the reference's val_step instead (eval mode, no_grad forward + CE; kfold_train_adversarial.py:144-161)")
    ap.add_argument("--no-fused-adam", action="store_true")
    ap.add_argument("--roofline-only", action="store_true",
                    help="skip the train-step timing: only the dominant-kernel loop (so that a rocprofv3 --stats run "
                         "of this command averages exactly the launches the roofline entry quotes)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step from a hipGraph (measured slower than eager on ROCm 7.2: 28.0 vs 16.8 ms)")
    args = ap.parse_args()

    from transmf_ad_amd import model_ad, ops, _lib
    from transmf_ad_amd.parallel import GradAllReduce, init_from_env

    rank, local, world = init_from_env()
    if world != args.gpus:
        if args.gpus != 1 or world != 1:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback in the product path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    ops.set_conv_precision(args.precision)
    if args.storage == "bf16" and args.precision != "bf16":
        raise SystemExit("--storage bf16 needs --precision bf16")
    ops.set_activation_storage(args.storage)
    torch.manual_seed(0)
    if args.model == "ad":
        net = model_ad(dim=128, depth=3, heads=4, dim_head=32, mlp_dim=512, dropout=0.).to(dev)
    elif args.model == "cnn":
        from transmf_ad_amd import model_CNN_ad
        net = model_CNN_ad(dim=128).to(dev)
    else:
        from transmf_ad_amd import model_single
        net = model_single(128).to(dev)
    if world == 1 and os.environ.get("TMF_DDP_FORCE", "0") == "1":     # overhead measurement of the N>1 machinery
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        net = GradAllReduce(net)
    if world > 1:
        net = GradAllReduce(net)
    # same update rule as the reference's getOptimizer (utils/utils.py:38-39: Adam, lr 1e-4, wd 0), multi-tensor form
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, capturable=args.graph and world == 1,
                           fused=not args.graph and not args.no_fused_adam)
    crit = nn.CrossEntropyLoss()
    B, S = args.batch, args.size
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    vol = tuple(args.shape) if args.shape else (S, S, S)
    mri = torch.rand((B, 1) + vol, device=dev, generator=g)
    pet = torch.rand((B, 1) + vol, device=dev, generator=g)
    label = (torch.arange(B, device=dev) % 2).long()
    ones = torch.ones(B, dtype=torch.int64, device=dev)
    zeros = torch.zeros(B, dtype=torch.int64, device=dev)

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
        opt.zero_grad()
        if args.model == "single":          # kfold_train_single.py:91-113: plain CE on model_single(MRI)
            loss = crit(net(mri), label)
        else:
            lo, dm, dp = net(mri, pet)
            loss = (crit(dm, ones) + crit(dp, zeros)) / 2 + crit(lo, label)
        loss.backward()
        opt.step()
        return loss

    mode = "eager"
    if args.graph:
        # same kernels, same order, same numerics — replayed from a hipGraph instead of re-dispatched by Python
        try:
            from transmf_ad_amd.graphs import GraphedTrainStep

            def loss_fn(out, lab):
                lo, dm, dp = out
                return (crit(dm, ones) + crit(dp, zeros)) / 2 + crit(lo, lab)

            graphed = GraphedTrainStep(net, opt, loss_fn, (mri, pet, label))
            eager_step = step

            def step():                      # noqa: F811
                return graphed(mri, pet, label)
            mode = "hipgraph"
        except Exception as e:               # capture unsupported: fall back to the eager step, and say so
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eager", file=sys.stderr)
            opt = torch.optim.Adam(net.parameters(), lr=1e-4, fused=True)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    loss = torch.zeros((), device=dev)
    if args.roofline_only:
        args.warmup, args.steps = 0, 0
    # The first ~30 steps of a process pay one-off host costs that are not part of a step: lazy code-object loading,
    # the caching allocator growing until buffers recycle (hipMalloc synchronises), hipBLASLt heuristics, and for N > 1
    # the RCCL communicator / channel set-up on the first collectives (measured: 11.4 ms/step over steps 6-25 against
    # 7.3 ms from step 30 on in the host-bound bf16 mode; 23 vs 17.8 ms with a 1-rank RCCL group).  They run here as
    # set-up, before the W warm-up steps the contract asks for; the GPU-bound fp32 step does not change with them.
    setup_steps = 30 if args.steps > 0 else 0
    for _ in range(setup_steps):
        step()
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = max(time.perf_counter() - t0, 1e-9)
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    final_loss = loss.item()
    assert final_loss == final_loss, "loss is NaN"

    ms_per_step = dt / max(args.steps, 1) * 1e3
    pairs_per_s = world * B * args.steps / dt

    # ---- roofline of the dominant kernel: conv3d_fwd_kernel (forward + data-gradient = 2/3 of
    # the MFMA work) at its largest launch, conv2.3 (32 -> 64 channels at (S/2)^3), timed live
    # with HIP events on the stream the kernel is launched on ----
    roof = None
    if rank == 0:
        s2 = S // 2
        x = torch.randn((B, s2, s2, s2, 32), device=dev)
        w = torch.randn((27, 32, 64), device=dev) * 0.03
        # The chip raises its clocks only after a few hundred ms of load: straight after start-up (--roofline-only)
        # this launch takes 0.87 ms, after the train steps above 0.78 ms.  Bring it to the loaded state with unrelated
        # work first, so the live number and a rocprofv3 --stats average of THIS kernel describe the same state.
        w_spin = torch.randn((27, 32, 32), device=dev) * 0.03          # conv2.0 shape: a different kernel instance
        t_spin = time.perf_counter()
        while time.perf_counter() - t_spin < float(os.environ.get("TMF_ROOF_SPIN_S", "0.5")):
            for _ in range(20):
                ops.conv3d_raw(x, w_spin, 32, 32, 3, True)
            torch.cuda.synchronize()
        del w_spin
        for _ in range(3):
            ops.conv3d_raw(x, w, 32, 64, 3, True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        z = torch.empty((B, s2, s2, s2, 64), device=dev)
        nblk = _lib.query("tmf_conv3d_stat_blocks", B, s2, s2, s2, 32, 64, 3)
        part = torch.empty((nblk, 2, 64), device=dev)
        st = torch.cuda.current_stream().cuda_stream
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            _lib.call("tmf_conv3d_fwd", x.data_ptr(), w.data_ptr(), z.data_ptr(), part.data_ptr(),
                      B, s2, s2, s2, 32, 64, 3, st)
        e1.record()
        torch.cuda.synchronize()
        k_ms = e0.elapsed_time(e1) / reps
        flops = 2.0 * 27 * 32 * 64 * B * s2 ** 3
        ach = flops / (k_ms * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": f"conv3d_fwd_kernel<FwdCfg<3,16,1,2,8,1,4,8,8,3>> @conv2.3 (B={B}, {s2}^3, 32->64 ch)", "achieved": round(ach, 2),
                "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4),
                "traffic": _profiled_traffic() if (B, S) == (8, 96) else None,
                "traffic_unit": "HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, "
                "profiles/r01_pmc_traffic_conv2.3_fwd.json)", "launch_ms": round(k_ms, 4), "flops_per_launch": flops}
        if args.precision == "bf16":
            # the opt-in bf16 mode's dominant kernel, priced against BOTH ceilings: it is far from the bf16 MFMA
            # peak by construction (fp32 activations in HBM: 384 B per voxel for 2*27*32*64 flops)
            wb = ops.pack_weight_bf16(torch.randn((64, 32, 3, 3, 3), device=dev) * 0.03)
            nb16 = _lib.query("tmf_conv3d_bf16_stat_blocks", B, s2, s2, s2, 64)
            part16 = torch.empty((nb16, 2, 64), device=dev)
            for _ in range(3):
                ops.conv3d_bf16_raw(x, wb, 32, 64, True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(reps):
                _lib.call("tmf_conv3d_fwd_bf16", x.data_ptr(), wb.data_ptr(), z.data_ptr(), part16.data_ptr(),
                          B, s2, s2, s2, 32, 64, st)
            e1.record()
            torch.cuda.synchronize()
            k16 = e0.elapsed_time(e1) / reps
            ach16 = flops / (k16 * 1e-3) / 1e12
            alg_bytes = (x.numel() + z.numel()) * 4.0
            roof = {"bound": "mfma", "kernel": f"conv3d_fwd_bf16_kernel<1> @conv2.3 (B={B}, {s2}^3, 32->64 ch)",
                    "achieved": round(ach16, 2), "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach16 / PEAK_BF16_MFMA_TFLOPS, 4), "traffic": None,
                    "launch_ms": round(k16, 4), "flops_per_launch": flops,
                    "hbm": {"algorithmic_bytes_per_launch": alg_bytes, "achieved_GBps": round(alg_bytes / k16 / 1e6, 1),
                            "peak_GBps": 8000.0, "frac": round(alg_bytes / k16 / 1e6 / 8000.0, 4)},
                    "fp32_kernel_same_shape": {"launch_ms": round(k_ms, 4), "achieved": round(ach, 2),
                                               "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4)}}
        del x, w, z, part
        # every 3x3x3 layer of one encoder, forward / data-gradient / weight-gradient launches (fp32 kernels, same
        # loaded clock state, 10 launches each): TFLOP/s and fraction of the fp32-MFMA peak
        if not args.roofline_only and args.precision == "fp32":
            def _t(fn, n=10):
                fn()
                torch.cuda.synchronize()
                a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a0.record()
                for _ in range(n):
                    fn()
                a1.record()
                torch.cuda.synchronize()
                return a0.elapsed_time(a1) / n
            layers = {}
            for name, ci, co, div in (("conv2.0", 32, 32, 2), ("conv2.3", 32, 64, 2), ("conv3.0", 64, 64, 4),
                                      ("conv3.3", 64, 128, 4), ("conv4.0", 128, 256, 8)):
                sl = S // div
                xl = torch.randn((B, sl, sl, sl, ci), device=dev)
                dzl = torch.randn((B, sl, sl, sl, co), device=dev)
                wl = torch.randn((27, ci, co), device=dev) * 0.03
                wdl = torch.randn((27, co, ci), device=dev) * 0.03
                fl = 2.0 * 27 * ci * co * B * sl ** 3
                row = {}
                for key, fn in (("fwd", lambda: ops.conv3d_raw(xl, wl, ci, co, 3, True)),
                                ("dgrad", lambda: ops.conv3d_raw(dzl, wdl, co, ci, 3, False)),
                                ("wgrad", lambda: ops.conv3d_wgrad(xl, dzl, ci, co, 3))):
                    ms = _t(fn)
                    row[key] = {"ms": round(ms, 4), "tflops": round(fl / ms / 1e9, 1),
                                "frac": round(fl / ms / 1e9 / PEAK_FP32_MFMA_TFLOPS, 3)}
                layers[name] = row
                del xl, dzl, wl, wdl
            roof["layers"] = layers

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import tmf_oracle as O         # test infrastructure, used ONLY as the timed CPU baseline
        # MKL-DNN conv3d scales poorly past a few dozen threads (256 SMT threads: 85 s/step vs ~8 s on 8): cap at 32
        sec, threads = O.cpu_train_step_seconds(args.cpu_batch, S, steps=3, warmup=1,
                                                threads=min(args.cpu_threads, os.cpu_count()))
        cpu_model = "unknown CPU"
        try:
            with open("/proc/cpuinfo") as f:
                for line in f:
                    if line.startswith("model name"):
                        cpu_model = line.split(":", 1)[1].strip()
                        break
        except OSError:
            pass
        cpu = {"value": round(args.cpu_batch / sec, 4), "unit": "volume-pairs/s", "cores": threads, "kind": "port",
               "sample": f"oracle model_ad train-mode fwd+bwd, batch {args.cpu_batch} of 1x{S}^3 pairs "
                         f"(1/{max(1, B // args.cpu_batch)} of one batch-{B} step), 1 warm-up + 3 timed steps, "
                         f"{sec:.2f} s/step, {cpu_model}, host cpu_count={os.cpu_count()}, torch threads={threads}"}

    if rank == 0:
        gf = conv_flops_per_pair(vol) * (0.5 if args.model == "single" else 1.0)
        if args.eval:       # forward only: sum of F_l per stream
            gf = gf * (32.219 / 95.125) if vol == (96, 96, 96) else gf / 3.0
        vtxt = f"{S}^3" if not args.shape else "x".join(map(str, vol))
        model_desc = {"ad": "model_ad(dim=128,depth=3,heads=4,dim_head=32,mlp_dim=512)",
                      "cnn": "model_CNN_ad(dim=128) [BASELINE configs[4], dual-modality reading of --model CNN]",
                      "single": "model_single(128), MRI only [BASELINE configs[4], single-modality reading]"}[args.model]
        out = {
            "metric": ((f"volume-pairs/sec fwd+bwd(+Adam), {vtxt} MRI+PET batch={B} per GPU" if args.model != "single"
                        else f"volumes/sec fwd+bwd(+Adam), {vtxt} MRI only batch={B} per GPU") if not args.eval else
                       f"volume-pairs/sec eval forward (val_step), {vtxt} batch={B} per GPU"),
            "value": round(pairs_per_s, 3), "unit": "volume-pairs/s" if args.model != "single" else "volumes/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
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
                       "step": "val_step: eval-mode no_grad forward + CE" if args.eval else "zero_grad+fwd+loss+bwd+allreduce+Adam",
                       "dispatch": mode,
                       "setup_steps_untimed": setup_steps},
            "conv_tflops_whole_step": round(pairs_per_s / world * gf / 1e12, 2),
            "loss": round(final_loss, 6),
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
