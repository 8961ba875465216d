"""Helpers shared by the tests: load a golden fixture, rebuild its inputs, run the oracle."""
import json
import os

import numpy as np
import torch

from oracle import params as P
from oracle import tmf_oracle as O

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# parameters whose gradient is mathematically zero (a constant shift ahead of a
# batch-statistics normalisation): the reference's own fp32 values there are
# rounding noise (SURVEY.md §7 "Hard parts"), so they are only bounded, not matched.
def zero_grad_keys(spec, model, depth=None):
    z = {k for k in spec if k.endswith(".bias") and len(spec[k[:-4] + "weight"][1]) == 5}   # conv biases
    if model in ("model_ad", "model_CNN_ad"):
        z.add("D.0.bias")
    if model == "model_ad":
        z |= {"fc_cls.0.bias", "fc_cls.4.bias"}
        last = max(int(k.split(".")[2]) for k in spec if k.startswith("fuse_transformer.layers."))
        z.add(f"fuse_transformer.layers.{last}.1.norm.bias")    # constant shift of cls ahead of BN1d
    return z


class Golden:
    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.meta = json.loads(bytes(self.z["meta"]).decode())
        self.model = self.meta["model"]
        self.kw = self.meta["kwargs"]
        self.size = tuple(self.meta["size"])
        self.batch = self.meta["batch"]
        if self.model == "model_ad":
            self.spec = O.state_spec(self.model, **self.kw)
        else:
            self.spec = O.state_spec(self.model, dim=self.kw["dim"])

    def has(self, key):
        return key in self.z.files

    def __getitem__(self, key):
        return self.z[key]

    def arrays(self):
        return P.init_arrays(self.spec, seed=self.meta["param_seed"])

    def inputs(self):
        return P.make_inputs(self.batch, self.size, seed=self.meta["input_seed"],
                             kind=self.meta.get("input_kind", "uniform"))

    def masks(self):
        return P.make_masks(self.batch, seed=self.meta["mask_seed"])

    def fusion_dropout(self):
        return float(self.meta.get("fusion_dropout", 0.0))

    def fusion_masks(self):
        """Scaled keep-masks (float32 arrays) of the fusion block's Dropout modules per Transformer instance, or None."""
        p = self.fusion_dropout()
        if p <= 0:
            return None
        tokens = (self.size[0] // 16) * (self.size[1] // 16) * (self.size[2] // 16)
        keep = P.make_fusion_masks(self.batch * tokens, 2 * self.kw["depth"], p, self.kw["dim"], self.kw["mlp_dim"],
                                   seed=self.meta["fusion_mask_seed"])
        return [tuple((k.astype(np.float32) / np.float32(1.0 - p)) for k in trip) for trip in keep]


def available(name):
    return os.path.exists(os.path.join(GOLDEN_DIR, name + ".npz"))


def probe(t):
    t = t.detach().double().reshape(-1).cpu()
    idx = torch.from_numpy(P.probe_indices(t.numel()))
    return np.concatenate([[t.mean().item(), t.std(unbiased=False).item()], t[idx].numpy()])


def gprobe(g):
    g = g.detach().double().reshape(-1).cpu()
    idx = torch.from_numpy(P.probe_indices(g.numel()))
    return np.concatenate([[g.sum().item(), g.abs().sum().item(), g.abs().max().item()], g[idx].numpy()])


def run_oracle(g: Golden, dtype=torch.float32, train=True, backward=True, keep_graph=False):
    """Run the oracle on a fixture's inputs.  Returns dict(outs, loss, probes, grads, state).
    keep_graph: leave the parameters requiring grad (and the probes attached) without calling backward."""
    S = O.to_state(g.arrays(), g.spec, dtype=dtype, requires_grad=train and (backward or keep_graph))
    mri, pet, y = g.inputs()
    mri, pet, y = torch.from_numpy(mri).to(dtype), torch.from_numpy(pet).to(dtype), torch.from_numpy(y)
    probes = {}
    res = {}
    ctx = torch.enable_grad() if train else torch.no_grad()
    with ctx:
        if g.model == "model_ad":
            k1, k2 = g.masks()
            fm = g.fusion_masks()
            if fm is not None:
                fm = [tuple(torch.from_numpy(k).to(dtype) for k in trip) for trip in fm]
            lo, dm, dp = O.model_ad_forward(S, mri, pet, dim=g.kw["dim"], depth=g.kw["depth"], heads=g.kw["heads"],
                                            train=train, dropout_masks=(torch.from_numpy(k1), torch.from_numpy(k2)),
                                            probes=probes, fusion_masks=fm)
            outs = dict(logits=lo, d_mri=dm, d_pet=dp)
            loss = O.adversarial_loss(lo, dm, dp, y)
        elif g.model == "model_CNN_ad":
            lo, dm, dp = O.model_cnn_ad_forward(S, mri, pet, dim=g.kw["dim"], train=train, probes=probes)
            outs = dict(logits=lo, d_mri=dm, d_pet=dp)
            loss = O.adversarial_loss(lo, dm, dp, y)
        else:
            lo = O.model_single_forward(S, mri, dim=g.kw["dim"], train=train, probes=probes)
            outs = dict(logits=lo)
            loss = torch.nn.functional.cross_entropy(lo, y)
        if train and backward:
            loss.backward()
            res["grads"] = O.grads_of(S, g.spec)
    res.update(outs=outs, loss=loss, probes=probes, state=S)
    return res
