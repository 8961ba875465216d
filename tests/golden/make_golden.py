#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING THE REFERENCE.

Runs only in the authoring container (needs /root/reference); its outputs — small
.npz files — are committed and are the only thing that travels.  Usage:

    python tests/golden/make_golden.py [case ...]      # default: all cases

Per case the reference module is built, loaded (strict) with the regenerable
parameters of oracle/params.py, and run on the regenerable volumes:
  * train mode, fp32 and fp64, with the two fc_cls Dropout(0.5) masks forced to
    oracle/params.make_masks() by forward hooks;  loss of
    kfold_train_adversarial.py:119-131, backward;
  * eval mode, fp32, no_grad (val_step, kfold_train_adversarial.py:144-161).
Stored: logits / D logits / loss, the 512-d cls vector, activation probes
(mean, std, 16 samples) after every conv block and every Transformer instance,
per-parameter gradient probes (sum, |sum|, max|.|, 16 samples), all BN buffers
after the train step, and the state_dict key/shape list.
"""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from oracle import params as P          # noqa: E402
from oracle import tmf_oracle as O      # noqa: E402

CASES = {
    # name: (model, ctor kwargs, volume size, batch, run fp64?)
    "ad_tiny":    ("model_ad", dict(dim=32, depth=2, heads=4, dim_head=8, mlp_dim=128), (32, 32, 32), 2, True),
    "ad_ragged":  ("model_ad", dict(dim=32, depth=2, heads=4, dim_head=8, mlp_dim=128), (35, 38, 33), 3, True),
    "ad_mid":     ("model_ad", dict(dim=128, depth=3, heads=4, dim_head=32, mlp_dim=512), (48, 48, 48), 2, True),
    "ad_full_b2": ("model_ad", dict(dim=128, depth=3, heads=4, dim_head=32, mlp_dim=512), (96, 96, 96), 2, True),
    "ad_full_b8": ("model_ad", dict(dim=128, depth=3, heads=4, dim_head=32, mlp_dim=512), (96, 96, 96), 8, True),
    # well-conditioned full-size case: structured volumes (oracle/params.make_inputs_blobs)
    "ad_full_b2_blobs": ("model_ad", dict(dim=128, depth=3, heads=4, dim_head=32, mlp_dim=512), (96, 96, 96), 2, True, "blobs"),
    # the reference's real ADNI volume shape (datasets/ADNI.py:93): odd pooled sizes 45x54x45 / 22x27x22 / 11x13x11
    "ad_adni_b2": ("model_ad", dict(dim=128, depth=3, heads=4, dim_head=32, mlp_dim=512), (91, 109, 91), 2, True, "blobs"),
    # BASELINE configs[2]: 128^3, batch 8 — fp32 reference run that the bf16 modes are gated against
    "ad_128_b8":  ("model_ad", dict(dim=128, depth=3, heads=4, dim_head=32, mlp_dim=512), (128, 128, 128), 8, False, "blobs"),
    # Dropout ACTIVE in the fusion block (options/option.py:39 --dropout; networks.py:131,133,153): the reference is built with
    # dropout=0.3 and every Transformer Dropout module is forced to the regenerable masks of oracle/params.make_fusion_masks
    "ad_mid_drop": ("model_ad", dict(dim=128, depth=3, heads=4, dim_head=32, mlp_dim=512), (48, 48, 48), 2, True, "blobs", 0.3),
    # the head geometry of the reference's OTHER entry script (train_adversarial.py:30-31: heads=8, dim_head=dim // 8)
    "ad_mid_h8": ("model_ad", dict(dim=128, depth=3, heads=8, dim_head=16, mlp_dim=512), (48, 48, 48), 2, True, "blobs"),
    "cnn_tiny":   ("model_CNN_ad", dict(dim=32), (32, 32, 32), 2, True),
    "cnn_mid":    ("model_CNN_ad", dict(dim=128), (48, 40, 48), 2, True),
    "single_mid": ("model_single", dict(dim=128), (48, 48, 48), 3, True),
    # BASELINE configs[4] at the full volume size (structured volumes, batch 2): the conv-only models against the reference
    "cnn_full_b2":    ("model_CNN_ad", dict(dim=128), (96, 96, 96), 2, True, "blobs"),
    "single_full_b2": ("model_single", dict(dim=128), (96, 96, 96), 2, True, "blobs"),
    # BASELINE configs[4] at its REAL batch (16, 96^3; structured volumes), both readings of "--model CNN": the batch the
    # bench lines of these models are quoted on, pinned against the reference itself (fp32 and fp64 runs)
    "cnn_full_b16":    ("model_CNN_ad", dict(dim=128), (96, 96, 96), 16, True, "blobs"),
    "single_full_b16": ("model_single", dict(dim=128), (96, 96, 96), 16, True, "blobs"),
}
DEFAULT = [c for c in CASES if c not in ("ad_full_b8", "ad_128_b8", "cnn_full_b16", "single_full_b16")]


def build_reference(model, kw, dropout=0.):
    from models.mymodel import model_ad, model_CNN_ad, model_single
    if model == "model_ad":
        return model_ad(dropout=dropout, **kw)
    if model == "model_CNN_ad":
        return model_CNN_ad(**kw)
    return model_single(kw["dim"])


def spec_for(model, kw):
    if model == "model_ad":
        return O.state_spec(model, **kw)
    return O.state_spec(model, dim=kw["dim"])


def probe(t):
    t = t.detach().double().reshape(-1)
    idx = torch.from_numpy(P.probe_indices(t.numel()))
    return np.concatenate([[t.mean().item(), t.std(unbiased=False).item()], t[idx].numpy()])


def gprobe(g):
    g = g.detach().double().reshape(-1)
    idx = torch.from_numpy(P.probe_indices(g.numel()))
    return np.concatenate([[g.sum().item(), g.abs().sum().item(), g.abs().max().item()], g[idx].numpy()])


def attach_probes(net, model, store):
    """Forward hooks giving the same probe points as oracle snet_forward/fusion_forward."""
    hs = []

    def hook(name):
        def f(_m, _i, o):
            store[name] = probe(o)
        return f

    cnns = {"model_single": ["cnn"]}.get(model, ["mri_cnn", "pet_cnn"])
    for c in cnns:
        s = getattr(net, c)
        pts = {"conv1.0": s.conv1, "conv2.0": s.conv2[2], "conv2.3": s.conv2, "conv3.0": s.conv3[2],
               "conv3.3": s.conv3, "conv4.0": s.conv4[2], "conv4.3": s.conv4}
        for k, m in pts.items():
            hs.append(m.register_forward_hook(hook(f"{c}.{k}")))
    if model == "model_ad":
        for l, pair in enumerate(net.fuse_transformer.layers):
            for s in (0, 1):
                hs.append(pair[s].register_forward_hook(hook(f"fuse_transformer.layers.{l}.{s}")))
        hs.append(net.fuse_transformer.register_forward_hook(hook("cls")))
    return hs


def run_case(name):
    model, kw, size, B, do64 = CASES[name][:5]
    kind = CASES[name][5] if len(CASES[name]) > 5 else "uniform"
    drop_p = CASES[name][6] if len(CASES[name]) > 6 else 0.0
    spec = spec_for(model, kw)
    arrays = P.init_arrays(spec, seed=7)
    mri, pet, y = P.make_inputs(B, size, seed=1234, kind=kind)
    k1, k2 = P.make_masks(B)
    out = {}
    meta = dict(case=name, model=model, kwargs=kw, size=list(size), batch=B, param_seed=7, input_seed=1234,
                mask_seed=99, input_kind=kind, torch=torch.__version__, fusion_dropout=drop_p, fusion_mask_seed=123,
                keys=[[k, list(s)] for k, (_kind, s) in spec.items()])

    for prec, dt in (("f32", torch.float32), ("f64", torch.float64)):
        if prec == "f64" and not do64:
            continue
        t0 = time.time()
        torch.manual_seed(0)
        net = build_reference(model, kw, drop_p)
        sd = net.state_dict()
        assert list(sd.keys()) == list(spec.keys()), "state_dict key order differs from oracle spec"
        for k, v in sd.items():
            assert tuple(v.shape) == tuple(spec[k][1]), (k, v.shape, spec[k][1])
        net.load_state_dict({k: torch.from_numpy(np.asarray(arrays[k])) for k in spec}, strict=True)
        net = net.to(dt)
        net.train()
        pr = {}
        hooks = attach_probes(net, model, pr)
        if model == "model_ad":
            m1 = torch.from_numpy(k1).to(dt)
            m2 = torch.from_numpy(k2).to(dt)
            hooks.append(net.fc_cls[3].register_forward_hook(lambda _m, i, _o: i[0] * m1 * 2.0))
            hooks.append(net.fc_cls[7].register_forward_hook(lambda _m, i, _o: i[0] * m2 * 2.0))
        if drop_p > 0:
            tokens = (size[0] // 16) * (size[1] // 16) * (size[2] // 16)
            fm = P.make_fusion_masks(B * tokens, 2 * kw["depth"], drop_p, kw["dim"], kw["mlp_dim"], seed=123)
            inst = 0
            for pair in net.fuse_transformer.layers:
                for tr in pair:
                    at, ff = tr.layers[0][0].fn, tr.layers[0][1].fn
                    for mod, keep in zip((at.to_out[1], ff.net[2], ff.net[4]), fm[inst]):
                        assert isinstance(mod, torch.nn.Dropout) and mod.p == drop_p
                        # scaled keep-mask as the fp32 value the device path is handed (also in the fp64 run)
                        mk = torch.from_numpy(keep.astype(np.float32) / np.float32(1.0 - drop_p)).to(dt)
                        hooks.append(mod.register_forward_hook(lambda _m, i, _o, mk=mk: i[0] * mk.reshape(i[0].shape)))
                    inst += 1
        xm, xp = torch.from_numpy(mri).to(dt), torch.from_numpy(pet).to(dt)
        yt = torch.from_numpy(y)
        crit = torch.nn.CrossEntropyLoss()
        if model == "model_single":
            lo = net(xm)
            loss = crit(lo, yt)
            outs = {"logits": lo}
        else:
            lo, dm, dp = net(xm, xp)
            ones = torch.ones(B, dtype=torch.int64)
            zeros = torch.zeros(B, dtype=torch.int64)
            loss = (crit(dm, ones) + crit(dp, zeros)) / 2 + crit(lo, yt)
            outs = {"logits": lo, "d_mri": dm, "d_pet": dp}
        loss.backward()
        for h in hooks:
            h.remove()
        for k, v in outs.items():
            out[f"{prec}/train/{k}"] = v.detach().double().numpy()
        out[f"{prec}/train/loss"] = np.float64(loss.item())
        for k, v in pr.items():
            out[f"{prec}/probe/{k}"] = v
        for k, p_ in net.named_parameters():
            g = p_.grad if p_.grad is not None else torch.zeros_like(p_)
            out[f"{prec}/grad/{k}"] = gprobe(g)
        if prec == "f32":
            for k, b in net.named_buffers():
                out[f"f32/buf/{k}"] = b.detach().double().numpy()
        print(f"  {name} {prec} train: loss={loss.item():.8f}  ({time.time() - t0:.1f}s)", flush=True)
        del net

    # eval mode (val_step): fresh buffers from the fixture, fp32
    net = build_reference(model, kw, drop_p)
    net.load_state_dict({k: torch.from_numpy(np.asarray(arrays[k])) for k in spec}, strict=True)
    net.eval()
    with torch.no_grad():
        xm, xp = torch.from_numpy(mri), torch.from_numpy(pet)
        if model == "model_single":
            out["f32/eval/logits"] = net(xm).double().numpy()
        else:
            lo, dm, dp = net(xm, xp)
            out["f32/eval/logits"] = lo.double().numpy()
            out["f32/eval/d_mri"] = dm.double().numpy()
            out["f32/eval/d_pet"] = dp.double().numpy()
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KB)")


if __name__ == "__main__":
    torch.set_num_threads(os.cpu_count())
    for c in (sys.argv[1:] or DEFAULT):
        run_case(c)
