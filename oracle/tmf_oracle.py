"""CPU oracle for the TransMF_AD forward/backward hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  The product package (``transmf_ad_amd``) never does, and it has no
CPU fallback: without the HIP library it raises.

What this file is: a *functional* restatement, on stock PyTorch CPU ops, of the
algorithm the reference implements in

    /root/reference/models/networks.py   (sNet :18-61, PreNorm :114-121,
                                          FeedForward :125-137, Attention :141-175,
                                          Transformer :215-230,
                                          CrossTransformer_MOD_AVG :255-281)
    /root/reference/models/mymodel.py    (model_single :13-37, model_CNN_ad :144-179,
                                          model_ad :182-222)
    /root/reference/models/gradient_reversal/functional.py :4-18
    /root/reference/kfold_train_adversarial.py :101-136 (train_step), :144-161 (val_step)
    /root/reference/kfold_train_single.py :91-113

It is written against a flat ``{state_dict key: tensor}`` dictionary whose keys
and shapes are exactly the reference's ``state_dict()`` (so checkpoints
interchange), but shares no code with the reference: the networks are driven by
the layer tables below.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the reference in
the authoring container, runs it on the regenerable inputs/parameters of
``oracle/params.py`` and commits the outputs under ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks this file against every one of those
vectors (the reference itself ships no tests or golden vectors — SURVEY.md §4).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

BN_EPS = 1e-5          # torch.nn.BatchNorm3d / BatchNorm1d default (networks.py:23)
BN_MOMENTUM = 0.1      # torch default
LN_EPS = 1e-5          # torch.nn.LayerNorm default (networks.py:117)
LRELU_SLOPE = 0.01     # torch.nn.LeakyReLU default (networks.py:24)
REVGRAD_ALPHA = 2.0    # mymodel.py:209


# --------------------------------------------------------------------------
# layer tables
# --------------------------------------------------------------------------

def snet_layers(dim: int) -> List[dict]:
    """The seven conv blocks of ``sNet`` (networks.py:21-53).

    Each entry: state_dict sub-prefix of the conv and of its BatchNorm3d,
    channel counts, kernel edge and the pool that follows the activation.
    """
    q, h, d, d2 = dim // 4, dim // 2, dim, dim * 2
    return [
        dict(conv="conv1.0", bn="conv1.1", cin=1,  cout=q,  k=3, pool="max"),   # :21-26
        dict(conv="conv2.0", bn="conv2.1", cin=q,  cout=q,  k=3, pool=None),    # :28-30
        dict(conv="conv2.3", bn="conv2.4", cin=q,  cout=h,  k=3, pool="max"),   # :31-34
        dict(conv="conv3.0", bn="conv3.1", cin=h,  cout=h,  k=3, pool=None),    # :37-39
        dict(conv="conv3.3", bn="conv3.4", cin=h,  cout=d,  k=3, pool="max"),   # :40-43
        dict(conv="conv4.0", bn="conv4.1", cin=d,  cout=d2, k=3, pool=None),    # :46-48
        dict(conv="conv4.3", bn="conv4.4", cin=d2, cout=d,  k=1, pool="avg"),   # :49-52
    ]


def _bn_entries(spec: "OrderedDict[str, tuple]", p: str, c: int) -> None:
    spec[p + ".weight"] = ("param", (c,))
    spec[p + ".bias"] = ("param", (c,))
    spec[p + ".running_mean"] = ("buffer", (c,))
    spec[p + ".running_var"] = ("buffer", (c,))
    spec[p + ".num_batches_tracked"] = ("buffer", ())


def _snet_spec(spec, pre: str, dim: int) -> None:
    for L in snet_layers(dim):
        k = L["k"]
        spec[f"{pre}{L['conv']}.weight"] = ("param", (L["cout"], L["cin"], k, k, k))
        spec[f"{pre}{L['conv']}.bias"] = ("param", (L["cout"],))
        _bn_entries(spec, f"{pre}{L['bn']}", L["cout"])


def _transformer_spec(spec, pre: str, dim: int, heads: int, dim_head: int, mlp_dim: int) -> None:
    """One ``Transformer(dim, depth=1, ...)`` instance (networks.py:215-224).

    Registration order in the reference: ``layers`` is assigned before ``norm``
    (networks.py:218-219), so ``layers.*`` keys precede ``norm.*``.
    """
    inner = heads * dim_head
    a = pre + "layers.0.0."      # PreNorm(Attention)
    f = pre + "layers.0.1."      # PreNorm(FeedForward)
    spec[a + "norm.weight"] = ("param", (dim,))
    spec[a + "norm.bias"] = ("param", (dim,))
    spec[a + "fn.to_q.weight"] = ("param", (inner, dim))
    spec[a + "fn.to_kv.weight"] = ("param", (2 * inner, dim))
    spec[a + "fn.to_out.0.weight"] = ("param", (dim, inner))
    spec[a + "fn.to_out.0.bias"] = ("param", (dim,))
    spec[f + "norm.weight"] = ("param", (dim,))
    spec[f + "norm.bias"] = ("param", (dim,))
    spec[f + "fn.net.0.weight"] = ("param", (mlp_dim, dim))
    spec[f + "fn.net.0.bias"] = ("param", (mlp_dim,))
    spec[f + "fn.net.3.weight"] = ("param", (dim, mlp_dim))
    spec[f + "fn.net.3.bias"] = ("param", (dim,))
    spec[pre + "norm.weight"] = ("param", (dim,))
    spec[pre + "norm.bias"] = ("param", (dim,))


def _linear_spec(spec, p: str, cin: int, cout: int) -> None:
    spec[p + ".weight"] = ("param", (cout, cin))
    spec[p + ".bias"] = ("param", (cout,))


def _disc_spec(spec, dim: int) -> None:
    """Discriminator ``D`` (mymodel.py:194): Linear-BN1d-ReLU-Linear."""
    _linear_spec(spec, "D.0", dim, 128)
    _bn_entries(spec, "D.1", 128)
    _linear_spec(spec, "D.3", 128, 2)


def state_spec(model: str, dim: int = 128, depth: int = 3, heads: int = 4,
               dim_head: int = 32, mlp_dim: int = 512) -> "OrderedDict[str, tuple]":
    """Ordered ``{key: (kind, shape)}`` equal to the reference ``state_dict()``.

    model: 'model_ad' (mymodel.py:182-194), 'model_CNN_ad' (:144-154),
    'model_single' (:13-20).  Order follows attribute registration order.
    """
    spec: "OrderedDict[str, tuple]" = OrderedDict()
    if model == "model_ad":
        _snet_spec(spec, "mri_cnn.", dim)
        _snet_spec(spec, "pet_cnn.", dim)
        for l in range(depth):
            for s in (0, 1):
                _transformer_spec(spec, f"fuse_transformer.layers.{l}.{s}.", dim, heads, dim_head, mlp_dim)
        _linear_spec(spec, "fc_cls.0", dim * 4, 512)
        _bn_entries(spec, "fc_cls.1", 512)
        _linear_spec(spec, "fc_cls.4", 512, 64)
        _bn_entries(spec, "fc_cls.5", 64)
        _linear_spec(spec, "fc_cls.8", 64, 2)
        _disc_spec(spec, dim)
    elif model == "model_CNN_ad":
        _snet_spec(spec, "mri_cnn.", dim)
        _snet_spec(spec, "pet_cnn.", dim)
        _linear_spec(spec, "fc_cls.0", dim * 2, 128)
        _linear_spec(spec, "fc_cls.2", 128, 2)
        _disc_spec(spec, dim)
    elif model == "model_single":
        _snet_spec(spec, "cnn.", dim)
        _linear_spec(spec, "fc.0", 128, 64)      # hard-coded 128 (mymodel.py:20)
        _linear_spec(spec, "fc.2", 64, 2)
    else:
        raise ValueError(model)
    return spec


# --------------------------------------------------------------------------
# primitive blocks
# --------------------------------------------------------------------------

def _batch_norm(S: Dict[str, Tensor], p: str, x: Tensor, train: bool) -> Tensor:
    """BatchNorm{1d,3d} with torch semantics: batch statistics + running update
    (biased var for normalisation, unbiased into running_var) in train mode,
    running statistics in eval mode; num_batches_tracked += 1 per train call."""
    if train:
        S[p + ".num_batches_tracked"] += 1
    return F.batch_norm(x, S[p + ".running_mean"], S[p + ".running_var"],
                        S[p + ".weight"], S[p + ".bias"], train, BN_MOMENTUM, BN_EPS)


# --------------------------------------------------------------------------
# The opt-in bf16 modes of the HIP path (BASELINE configs[2]), restated: the SAME algorithm with the product path's
# rounding points made explicit, so that the bf16 kernels have an oracle of their own (tests/test_gpu_model.py).
#   conv_mode "bf16" : every 3x3x3 convolution product — forward, data gradient, weight gradient, incl. the first
#                      block — takes its two operands rounded to bf16 (RNE) and accumulates exactly (here: in the run's
#                      dtype, fp64 in the tests); the 1x1x1 layer stays exact.
#   conv_mode "bf16s": additionally the tensors BETWEEN the 3x3x3 blocks are bf16: the raw conv output z is rounded once
#                      (BatchNorm statistics still come from the unrounded accumulators), a block's output is rounded
#                      when the next block also runs on the bf16 kernels, and the gradients of those tensors (dz, and the
#                      data gradient written into a bf16 tensor) are rounded the same way.
# --------------------------------------------------------------------------

def _rb(t: Tensor) -> Tensor:
    """round to bf16 (RNE, from the fp32 value — what the kernels see) and back to the run's dtype"""
    return t.float().bfloat16().to(t.dtype)


class _ConvBf16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, pad):
        xr, wr = _rb(x), _rb(w)
        ctx.save_for_backward(xr, wr)
        ctx.pad = pad
        return F.conv3d(xr, wr, None, stride=1, padding=pad)

    @staticmethod
    def backward(ctx, dz):
        xr, wr = ctx.saved_tensors
        dzr = _rb(dz)
        dx = torch.nn.grad.conv3d_input(xr.shape, wr, dzr, stride=1, padding=ctx.pad) if ctx.needs_input_grad[0] else None
        dw = torch.nn.grad.conv3d_weight(xr, wr.shape, dzr, stride=1, padding=ctx.pad)
        return dx, dw, None


class _RoundSTE(torch.autograd.Function):
    """a tensor stored as bf16: the value is rounded on the way forward, its gradient on the way back"""

    @staticmethod
    def forward(ctx, x):
        return _rb(x)

    @staticmethod
    def backward(ctx, g):
        return _rb(g)


class _BnStoredZ(torch.autograd.Function):
    """Train-mode BatchNorm whose statistics come from the UNROUNDED conv output while the normalised tensor is the
    stored (bf16) one; backward is the kernels' closed form on the stored tensor with the saved statistics:
        xhat = (zs - mean) * invstd;  dgamma = sum(dy * xhat);  dbeta = sum(dy);
        dz = gamma * invstd * (dy - mean(dy) - xhat * mean(dy * xhat))"""

    @staticmethod
    def forward(ctx, zs, mean, invstd, gamma, beta):
        shp = (1, -1, 1, 1, 1)
        xhat = (zs - mean.view(shp)) * invstd.view(shp)
        ctx.save_for_backward(xhat, invstd, gamma)
        return xhat * gamma.view(shp) + beta.view(shp)

    @staticmethod
    def backward(ctx, dy):
        xhat, invstd, gamma = ctx.saved_tensors
        shp = (1, -1, 1, 1, 1)
        dims = (0, 2, 3, 4)
        n = dy.numel() / dy.shape[1]
        dbeta = dy.sum(dims)
        dgamma = (dy * xhat).sum(dims)
        dz = (gamma * invstd).view(shp) * (dy - (dbeta / n).view(shp) - xhat * (dgamma / n).view(shp))
        return dz, None, None, dgamma, dbeta


def _bf16_capable(L: dict) -> bool:
    return L["k"] == 3 and L["cin"] > 1 and L["cin"] % 8 == 0


def snet_forward(S: Dict[str, Tensor], pre: str, dim: int, x: Tensor, train: bool,
                 probes: Optional[dict] = None, conv_mode: str = "exact") -> Tensor:
    """``sNet.forward`` (networks.py:55-61).  x: (B,1,D,H,W) -> (B,dim,D/16,H/16,W/16).
    conv_mode: "exact" (the reference), or the restated bf16 modes of the HIP path (see above; train mode only)."""
    if conv_mode != "exact":
        return _snet_forward_bf16(S, pre, dim, x, train, probes, conv_mode)
    for L in snet_layers(dim):
        w, b = S[f"{pre}{L['conv']}.weight"], S[f"{pre}{L['conv']}.bias"]
        x = F.conv3d(x, w, b, stride=1, padding=1 if L["k"] == 3 else 0)
        x = _batch_norm(S, f"{pre}{L['bn']}", x, train)
        x = F.leaky_relu(x, LRELU_SLOPE)
        if L["pool"] == "max":
            x = F.max_pool3d(x, 2, 2)
        elif L["pool"] == "avg":
            x = F.avg_pool3d(x, 2, 2)
        if probes is not None:
            probes[f"{pre}{L['conv']}"] = x
    return x


def _snet_forward_bf16(S, pre, dim, x, train, probes, conv_mode):
    assert train and conv_mode in ("bf16", "bf16s")
    store16 = conv_mode == "bf16s"
    layers = snet_layers(dim)
    for i, L in enumerate(layers):
        w, b = S[f"{pre}{L['conv']}.weight"], S[f"{pre}{L['conv']}.bias"]
        bn = f"{pre}{L['bn']}"
        on_bf16 = L["cin"] == 1 or _bf16_capable(L)           # the fused first block has bf16 passes of its own
        if on_bf16:
            z = _ConvBf16.apply(x, w, 1)
        else:
            z = F.conv3d(x, w, None, stride=1, padding=1 if L["k"] == 3 else 0)
        # batch statistics from the unrounded accumulators; the conv bias only shifts the mean (folded: it reaches
        # running_mean, never the activations)
        dims = (0, 2, 3, 4)
        mean = z.mean(dims)
        var = z.var(dims, unbiased=False)
        n = z.numel() / z.shape[1]
        with torch.no_grad():
            S[bn + ".num_batches_tracked"] += 1
            S[bn + ".running_mean"].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * (mean + b).to(S[bn + ".running_mean"].dtype))
            S[bn + ".running_var"].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * (var * n / (n - 1)).to(S[bn + ".running_var"].dtype))
        invstd = (var + BN_EPS).rsqrt()
        z16 = store16 and _bf16_capable(L)                    # the first block never stores z
        if z16:
            y = _BnStoredZ.apply(_RoundSTE.apply(z), mean.detach(), invstd.detach(), S[bn + ".weight"], S[bn + ".bias"])
        else:
            shp = (1, -1, 1, 1, 1)
            y = (z - mean.view(shp)) * invstd.view(shp) * S[bn + ".weight"].view(shp) + S[bn + ".bias"].view(shp)
        y = F.leaky_relu(y, LRELU_SLOPE)
        if L["pool"] == "max":
            y = F.max_pool3d(y, 2, 2)
        elif L["pool"] == "avg":
            y = F.avg_pool3d(y, 2, 2)
        if store16 and i + 1 < len(layers) and on_bf16 and _bf16_capable(layers[i + 1]):
            y = _RoundSTE.apply(y)
        if probes is not None:
            probes[f"{pre}{L['conv']}"] = y
        x = y
    return x


def attention_forward(S, p: str, x: Tensor, ctx: Tensor, heads: int, mask_o: Optional[Tensor] = None) -> Tensor:
    """``Attention.forward`` (networks.py:157-175); p ends with 'fn.'.

    q from x (already layer-normed by the caller), k/v from the *raw* context.
    """
    B, N, _ = x.shape
    M = ctx.shape[1]
    q = F.linear(x, S[p + "to_q.weight"])
    kv = F.linear(ctx, S[p + "to_kv.weight"])
    inner = q.shape[-1]
    dh = inner // heads
    k, v = kv[..., :inner], kv[..., inner:]
    q = q.reshape(B, N, heads, dh).permute(0, 2, 1, 3)
    k = k.reshape(B, M, heads, dh).permute(0, 2, 1, 3)
    v = v.reshape(B, M, heads, dh).permute(0, 2, 1, 3)
    dots = torch.matmul(q, k.transpose(-1, -2)) * (dh ** -0.5)       # :169
    attn = torch.softmax(dots, dim=-1)                                # :171
    out = torch.matmul(attn, v)                                       # :173
    out = out.permute(0, 2, 1, 3).reshape(B, N, inner)
    y = F.linear(out, S[p + "to_out.0.weight"], S[p + "to_out.0.bias"])
    return y if mask_o is None else y * mask_o.reshape(y.shape).to(y.dtype)      # to_out.1 = Dropout (:153), scaled keep-mask


def transformer_forward(S, p: str, x: Tensor, ctx: Tensor, heads: int, masks=None) -> Tensor:
    """``Transformer(depth=1).forward`` (networks.py:226-230) with PreNorm
    (:120-121: only x is normed, the context passes through untouched).
    masks: None (Dropout inactive: p = 0 or eval) or the three SCALED keep-masks of the instance's Dropout modules
    (after to_out :153, after GELU :131, after the second Linear :133)."""
    a, f = p + "layers.0.0.", p + "layers.0.1."
    D = x.shape[-1]
    mo, mg, mf = masks if masks is not None else (None, None, None)
    xn = F.layer_norm(x, (D,), S[a + "norm.weight"], S[a + "norm.bias"], LN_EPS)
    x = attention_forward(S, a + "fn.", xn, ctx, heads, mo) + x
    xn = F.layer_norm(x, (D,), S[f + "norm.weight"], S[f + "norm.bias"], LN_EPS)
    h = F.gelu(F.linear(xn, S[f + "fn.net.0.weight"], S[f + "fn.net.0.bias"]))   # erf GELU (:130)
    if mg is not None:
        h = h * mg.reshape(h.shape).to(h.dtype)
    y = F.linear(h, S[f + "fn.net.3.weight"], S[f + "fn.net.3.bias"])
    if mf is not None:
        y = y * mf.reshape(y.shape).to(y.dtype)
    x = y + x
    return F.layer_norm(x, (D,), S[p + "norm.weight"], S[p + "norm.bias"], LN_EPS)


def fusion_forward(S, p: str, mri: Tensor, pet: Tensor, depth: int, heads: int,
                   probes: Optional[dict] = None, masks=None) -> Tensor:
    """``CrossTransformer_MOD_AVG.forward`` (networks.py:272-281) -> (B, 4*dim).
    masks: per Transformer instance in execution order (2 * depth entries) the three scaled Dropout keep-masks, or None."""
    for l in range(depth):
        tm = transformer_forward(S, f"{p}layers.{l}.0.", mri, pet, heads, None if masks is None else masks[2 * l])
        mri = tm + mri                                                                # :274
        tp = transformer_forward(S, f"{p}layers.{l}.1.", pet, mri, heads,             # uses the NEW mri
                                 None if masks is None else masks[2 * l + 1])
        pet = tp + pet                                                                # :275
        if probes is not None:          # Transformer-instance outputs (before the outer residual)
            probes[f"{p}layers.{l}.0"] = tm
            probes[f"{p}layers.{l}.1"] = tp
    return torch.cat([mri.mean(dim=1), pet.mean(dim=1),
                      mri.max(dim=1).values, pet.max(dim=1).values], dim=1)          # :276-281


class _RevGrad(torch.autograd.Function):
    """gradient_reversal/functional.py:4-18: identity forward, -alpha*g backward."""

    @staticmethod
    def forward(ctx, x, alpha):
        ctx.alpha = alpha
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return -ctx.alpha * g, None


def _disc_forward(S, v: Tensor, train: bool) -> Tensor:
    """``D`` (mymodel.py:194) applied to one modality's pooled vector."""
    h = F.linear(v, S["D.0.weight"], S["D.0.bias"])
    h = F.relu(_batch_norm(S, "D.1", h, train))
    return F.linear(h, S["D.3.weight"], S["D.3.bias"])


def _drop(x: Tensor, mask: Optional[Tensor], train: bool, p: float = 0.5) -> Tensor:
    """nn.Dropout(0.5) of fc_cls (mymodel.py:190-191).  ``mask`` (bool, keep=True)
    replaces the RNG so that train-mode outputs are reproducible."""
    if not train:
        return x
    if mask is None:
        return F.dropout(x, p, True)
    return x * mask.to(x.dtype) / (1.0 - p)


# --------------------------------------------------------------------------
# whole models
# --------------------------------------------------------------------------

def model_ad_forward(S, mri: Tensor, pet: Tensor, *, dim=128, depth=3, heads=4, train=True,
                     dropout_masks: Optional[Tuple[Tensor, Tensor]] = None,
                     probes: Optional[dict] = None, conv_mode: str = "exact", fusion_masks=None):
    """``model_ad.forward`` (mymodel.py:204-222) -> (logits, D_MRI_logits, D_PET_logits)."""
    m = snet_forward(S, "mri_cnn.", dim, mri, train, probes, conv_mode)
    p = snet_forward(S, "pet_cnn.", dim, pet, train, probes, conv_mode)
    vm = _RevGrad.apply(m.mean(dim=(2, 3, 4)), REVGRAD_ALPHA)          # :210
    vp = _RevGrad.apply(p.mean(dim=(2, 3, 4)), REVGRAD_ALPHA)          # :211
    d_m = _disc_forward(S, vm, train)                                   # :214
    d_p = _disc_forward(S, vp, train)                                   # :215
    B, C = m.shape[:2]
    mt = m.reshape(B, C, -1).transpose(1, 2)                            # 'b d x y z -> b (x y z) d' :218
    pt = p.reshape(B, C, -1).transpose(1, 2)
    cls = fusion_forward(S, "fuse_transformer.", mt, pt, depth, heads, probes, fusion_masks if train else None)
    if probes is not None:
        probes["cls"] = cls
    return fc_cls_forward(S, cls, train, dropout_masks), d_m, d_p


def fc_cls_forward(S, cls: Tensor, train: bool, dropout_masks=None) -> Tensor:
    """``model_ad.fc_cls`` (mymodel.py:190-192): Linear-BN1d-ReLU-Dropout(.5) x2 - Linear."""
    k1, k2 = dropout_masks if dropout_masks is not None else (None, None)
    h = F.linear(cls, S["fc_cls.0.weight"], S["fc_cls.0.bias"])
    h = _drop(F.relu(_batch_norm(S, "fc_cls.1", h, train)), k1, train)
    h = F.linear(h, S["fc_cls.4.weight"], S["fc_cls.4.bias"])
    h = _drop(F.relu(_batch_norm(S, "fc_cls.5", h, train)), k2, train)
    return F.linear(h, S["fc_cls.8.weight"], S["fc_cls.8.bias"])


def model_cnn_ad_forward(S, mri: Tensor, pet: Tensor, *, dim=128, train=True, probes=None):
    """``model_CNN_ad.forward`` (mymodel.py:162-179)."""
    m = snet_forward(S, "mri_cnn.", dim, mri, train, probes)
    p = snet_forward(S, "pet_cnn.", dim, pet, train, probes)
    gm, gp = m.mean(dim=(2, 3, 4)), p.mean(dim=(2, 3, 4))
    d_m = _disc_forward(S, _RevGrad.apply(gm, REVGRAD_ALPHA), train)
    d_p = _disc_forward(S, _RevGrad.apply(gp, REVGRAD_ALPHA), train)
    h = F.relu(F.linear(torch.cat([gm, gp], dim=1), S["fc_cls.0.weight"], S["fc_cls.0.bias"]))
    return F.linear(h, S["fc_cls.2.weight"], S["fc_cls.2.bias"]), d_m, d_p


def model_single_forward(S, img: Tensor, *, dim=128, train=True, probes=None):
    """``model_single.forward`` (mymodel.py:30-37)."""
    f = snet_forward(S, "cnn.", dim, img, train, probes).mean(dim=(2, 3, 4))
    h = F.relu(F.linear(f, S["fc.0.weight"], S["fc.0.bias"]))
    return F.linear(h, S["fc.2.weight"], S["fc.2.bias"])


def adversarial_loss(logits: Tensor, d_m: Tensor, d_p: Tensor, label: Tensor) -> Tensor:
    """kfold_train_adversarial.py:119-131: CE(logits,y) + (CE(D_mri,1) + CE(D_pet,0)) / 2."""
    ones = torch.ones(d_m.shape[0], dtype=torch.int64)
    zeros = torch.zeros(d_p.shape[0], dtype=torch.int64)
    ce = F.cross_entropy(logits, label)
    ad = (F.cross_entropy(d_m, ones) + F.cross_entropy(d_p, zeros)) / 2
    return ad + ce


# --------------------------------------------------------------------------
# state helpers
# --------------------------------------------------------------------------

def to_state(arrays: Dict[str, "object"], spec, dtype=torch.float32, requires_grad=True) -> Dict[str, Tensor]:
    """numpy arrays (oracle/params.py) -> tensor state; params get requires_grad."""
    S: Dict[str, Tensor] = {}
    for k, (kind, _shape) in spec.items():
        a = arrays[k]
        if k.endswith("num_batches_tracked"):
            S[k] = torch.as_tensor(a, dtype=torch.int64).clone()
            continue
        t = torch.as_tensor(a).to(dtype).clone()
        if kind == "param" and requires_grad:
            t.requires_grad_(True)
        S[k] = t
    return S


def grads_of(S: Dict[str, Tensor], spec) -> Dict[str, Tensor]:
    out = {}
    for k, (kind, _s) in spec.items():
        if kind == "param":
            g = S[k].grad
            out[k] = torch.zeros_like(S[k]) if g is None else g
    return out


def cpu_train_step_seconds(batch: int, size, steps: int = 1, warmup: int = 1, seed: int = 1234,
                           threads: Optional[int] = None, model: str = "model_ad") -> Tuple[float, int]:
    """Time ``steps`` train-mode fwd+bwd passes of the reference step on the host cores (the ``cpu_baseline`` leg of
    bench.py): ``model_ad(128,3,4,32,512)`` / ``model_CNN_ad(128)`` with the adversarial loss
    (kfold_train_adversarial.py:101-136), or ``model_single(128)`` with plain CE (kfold_train_single.py:91-113).
    size: edge length or (D, H, W).  Returns (seconds per step, threads)."""
    import time
    import numpy as np
    from . import params as P
    if threads:
        torch.set_num_threads(threads)
    spec = state_spec(model)
    S = to_state(P.init_arrays(spec, seed=7), spec)
    vol = (size, size, size) if isinstance(size, int) else tuple(size)
    mri, pet, y = P.make_inputs(batch, vol, seed)
    mri, pet, y = torch.from_numpy(mri), torch.from_numpy(pet), torch.from_numpy(y)
    dt = []
    for it in range(warmup + steps):
        for k, (kind, _s) in spec.items():
            if kind == "param":
                S[k].grad = None
        t0 = time.perf_counter()
        if model == "model_single":
            F.cross_entropy(model_single_forward(S, mri, train=True), y).backward()
        else:
            fwd = model_ad_forward if model == "model_ad" else model_cnn_ad_forward
            lo, dm, dp = fwd(S, mri, pet, train=True)
            adversarial_loss(lo, dm, dp, y).backward()
        t1 = time.perf_counter()
        if it >= warmup:
            dt.append(t1 - t0)
    return float(np.mean(dt)), torch.get_num_threads()
