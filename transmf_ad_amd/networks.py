"""MI355X-native network blocks with the reference's constructor / forward signatures and
state_dict keys (reference: models/networks.py).

Each class keeps stock ``nn.Conv3d`` / ``nn.BatchNorm3d`` / ``nn.Linear`` / ``nn.LayerNorm``
sub-modules at the reference's attribute paths purely as PARAMETER HOLDERS — that is what makes
``state_dict()``, ``load_state_dict(strict=True)``, ``.to(device)`` and init loops such as
``isinstance(m, nn.Conv3d)`` behave exactly like the reference.  Their own ``forward`` is
never called: the compute goes through the HIP kernels in ``transmf_ad_amd.ops``.

Internal activation layout is channels-last (B, D, H, W, C); ``sNet.forward`` returns the
reference's (B, C, D, H, W) *view* of it (no copy); flattening that view to tokens
(mymodel.py:218 'b d x y z -> b (x y z) d') is then free as well.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn

from . import ops


def device_guard(forward):
    """Run a module's forward with its first tensor's GPU as the current HIP device: the kernels are launched on the
    CURRENT device's current stream, so a model moved with ``.to('cuda:1')`` works while device 0 is current (as the
    reference's stock modules do).  Backward needs no guard: autograd's device threads set the device themselves."""
    import functools

    @functools.wraps(forward)
    def wrapped(self, *args, **kwargs):
        x = next((a for a in args if isinstance(a, torch.Tensor)), None)
        if x is None:                        # keyword-only calls: net(mri=a, pet=b), attn(x=t)
            x = next((a for a in kwargs.values() if isinstance(a, torch.Tensor)), None)
        if x is not None and x.is_cuda and x.device.index != torch._C._cuda_getDevice():
            with torch.cuda.device(x.device):
                return forward(self, *args, **kwargs)
        return forward(self, *args, **kwargs)
    return wrapped


# ---------------------------------------------------------------------------------------
# sNet                                                         reference: networks.py:18-61
# ---------------------------------------------------------------------------------------

def _conv_block(cin, cout, k):
    return [nn.Conv3d(cin, cout, kernel_size=(k, k, k), padding=k // 2), nn.BatchNorm3d(cout), nn.LeakyReLU()]


class sNet(nn.Module):
    """Seven Conv3d-BatchNorm3d-LeakyReLU blocks, 2x2x2 pools after blocks 1, 3, 5 (max) and 7 (avg).

    forward(vol: (B, 1, D, H, W)) -> (B, dim, D//16, H//16, W//16)
    """

    # (sequential name, index of the conv inside it, pool that follows the block)
    _PLAN = (("conv1", 0, "max"), ("conv2", 0, None), ("conv2", 3, "max"), ("conv3", 0, None),
             ("conv3", 3, "max"), ("conv4", 0, None), ("conv4", 3, "avg"))

    def __init__(self, dim) -> None:
        super().__init__()
        q, h = dim // 4, dim // 2
        self.conv1 = nn.Sequential(*_conv_block(1, q, 3), nn.MaxPool3d(2, stride=2))
        self.conv2 = nn.Sequential(*_conv_block(q, q, 3), *_conv_block(q, h, 3), nn.MaxPool3d(2, stride=2))
        self.conv3 = nn.Sequential(*_conv_block(h, h, 3), *_conv_block(h, dim, 3), nn.MaxPool3d(2, stride=2))
        self.conv4 = nn.Sequential(*_conv_block(dim, dim * 2, 3), *_conv_block(dim * 2, dim, 1),
                                   nn.AvgPool3d(2, stride=2))
        # data-parallel bucketing hint (parallel.GradAllReduce): the gradients of conv3 / conv4 — 95 % of the encoder's bytes —
        # are complete well before the encoder's backward returns (tmf_snet_grads.deep_event); they get buckets of their own
        for name, p in self.named_parameters():
            p.tmf_bucket_group = ("sNet deep", id(self)) if name.startswith(("conv3", "conv4")) else ("sNet shallow",)
        self.tmf_precision = None           # None: follow the process default (ops.set_conv_precision / set_activation_storage)
        self.tmf_alone = False              # True: nothing runs beside this encoder (model_single): tmf_snet_desc.flags = TMF_SNET_ALONE
        self.tmf_algo = None                # None: the process options; a dict: this encoder's own kernels (set_algorithm)

    def set_precision(self, conv="fp32", storage="fp32"):
        """Precision of THIS encoder's convolution products ("fp32" | "bf16" | "fp32x") and of the activations between its
        blocks ("fp32" | "bf16", the latter with conv "bf16" only) — independent of any other module in the process;
        conv=None returns to the process default."""
        self.tmf_precision = None if conv is None else ops.make_precision(conv, storage)
        return self

    def set_algorithm(self, **algo):
        """Kernels of THIS encoder's fp32 convolutions in its whole-pass calls (tmf_snet_desc.flags, TMF_SNET_ALGO): conv_wino 0..3,
        wino_p, wino_x 0 | 1, c1_gram 0 | 1 | 2 (2: in the bf16 mode as well) — independent of tmf_set_option and of any other module; no arguments: the process options."""
        ops.snet_algo_flags(algo)            # (validates)
        self.tmf_algo = dict(algo) or None
        return self

    @device_guard
    def forward_channels_last(self, vol):
        if vol.dim() != 5 or vol.shape[1] != 1:
            raise ValueError(f"sNet expects (B, 1, D, H, W), got {tuple(vol.shape)}")
        B, _, D, H, W = vol.shape
        x = vol.reshape(B, D, H, W, 1)          # C == 1: NCDHW and NDHWC are the same bytes
        if self.training:                       # all seven counters in one multi-tensor launch
            nbt = [getattr(self, n)[i + 1].num_batches_tracked for n, i, _ in self._PLAN
                   if getattr(self, n)[i + 1].track_running_stats]
            if nbt:
                torch._foreach_add_(nbt, 1)
        blocks = [(getattr(self, n)[i], getattr(self, n)[i + 1], getattr(self, n)[i + 2]) for n, i, _ in self._PLAN]
        prec = ops.resolve_precision(self.tmf_precision)
        if (not self.training and not torch.is_grad_enabled() and prec[0] in ("fp32", "bf16") and ops.FUSE_EVAL_BLOCKS
                and all(bn.track_running_stats for _c, bn, _a in blocks)
                and self._one_call_ok(vol, blocks, eval_mode=True, prec=prec)):
            return ops.snet_eval_one_call(
                vol, blocks[-1][0].out_channels, tuple(float(bn.eps) for _c, bn, _a in blocks),
                tuple(float(a.negative_slope) for _c, _b, a in blocks),
                [(c.weight, c.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var) for c, bn, _a in blocks], prec,
                getattr(self, "tmf_algo", None))
        if self._one_call_ok(vol, blocks, prec=prec):
            params, buffers = [], []
            for conv, bn, _act in blocks:
                params += [conv.weight, conv.bias, bn.weight, bn.bias]
                buffers.append((bn.running_mean, bn.running_var) if bn.track_running_stats else (None, None))
            cfg = (blocks[-1][0].out_channels, tuple(float(bn.momentum) for _c, bn, _a in blocks),
                   tuple(float(bn.eps) for _c, bn, _a in blocks), tuple(float(a.negative_slope) for _c, _b, a in blocks), prec,
                   bool(getattr(self, "tmf_alone", False)), getattr(self, "tmf_algo", None))
            return ops.SNetTrain.apply(vol, cfg, buffers, *params)
        store16 = prec[1]
        for n_blk, (seq_name, i, pool) in enumerate(self._PLAN):
            seq = getattr(self, seq_name)
            conv, bn, act = seq[i], seq[i + 1], seq[i + 2]
            # bf16 activation storage: a block hands a bf16 tensor to the next block iff BOTH run on the bf16 kernels
            out16 = False
            if store16 and n_blk + 1 < len(self._PLAN):
                nxt = getattr(self, self._PLAN[n_blk + 1][0])[self._PLAN[n_blk + 1][1]]
                mine_ok = conv.in_channels == 1 or ops.bf16_conv_capable(conv.in_channels, conv.kernel_size[0])
                out16 = mine_ok and ops.bf16_conv_capable(nxt.in_channels, nxt.kernel_size[0])
            momentum = bn.momentum
            if momentum is None:                # BatchNorm(momentum=None): cumulative moving average 1 / n
                momentum = (1.0 / float(bn.num_batches_tracked.item())
                            if self.training and bn.track_running_stats else 0.0)
            x = ops.conv_bn_act_pool(x, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean,
                                     bn.running_var, self.training or not bn.track_running_stats,
                                     momentum=momentum, eps=bn.eps, slope=act.negative_slope, pool=pool,
                                     out_bf16=out16, precision=prec)
        return x                                 # (B, d, h, w, dim)

    def _one_call_ok(self, vol, blocks, eval_mode=False, prec=None):
        """Train-mode batch statistics in every block, the standard sNet(dim) geometry, no gradient wanted for the
        input: the whole pass is one library call (ops.SNetTrain); anything else goes block by block."""
        if vol.requires_grad or not vol.is_cuda or vol.dtype != torch.float32:
            return False
        B, _, D, H, W = vol.shape
        dim = blocks[-1][0].out_channels
        if not ops.snet_one_call_supported(B, D, H, W, dim, prec):
            return False
        q, h = dim // 4, dim // 2
        want = ((1, q, 3), (q, q, 3), (q, h, 3), (h, h, 3), (h, dim, 3), (dim, 2 * dim, 3), (2 * dim, dim, 1))
        for (conv, bn, _act), (ci, co, k) in zip(blocks, want):
            if (conv.in_channels, conv.out_channels, conv.kernel_size) != (ci, co, (k, k, k)):
                return False
            if not bn.affine or (not eval_mode and (not (self.training or not bn.track_running_stats) or bn.momentum is None)):
                return False
        return True

    def forward(self, mri):
        return self.forward_channels_last(mri).permute(0, 4, 1, 2, 3)


# ---------------------------------------------------------------------------------------
# transformer blocks                                    reference: networks.py:114-175, 215-230
# ---------------------------------------------------------------------------------------

class PreNorm(nn.Module):
    """LayerNorm on x only; keyword arguments (the attention context) pass through un-normalised."""

    def __init__(self, dim, fn):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.fn = fn

    @device_guard
    def forward(self, x, **kwargs):
        return self.fn(ops.layer_norm(x, self.norm.weight, self.norm.bias, self.norm.eps), **kwargs)


class FeedForward(nn.Module):
    def __init__(self, dim, hidden_dim, dropout=0.):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(dim, hidden_dim), nn.GELU(), nn.Dropout(dropout),
                                 nn.Linear(hidden_dim, dim), nn.Dropout(dropout))

    def forward(self, x):
        return self.net(x)


class Attention(nn.Module):
    """Multi-head attention; q from x, k/v from ``context`` (default x).  QK^T, softmax and AV run
    as one fused HIP kernel straight on the to_q / to_kv outputs."""

    def __init__(self, dim, heads=4, dim_head=64, dropout=0.):
        super().__init__()
        inner_dim = dim_head * heads
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.attend = nn.Softmax(dim=-1)          # kept for attribute parity; has no parameters
        self.to_q = nn.Linear(dim, inner_dim, bias=False)
        self.to_kv = nn.Linear(dim, inner_dim * 2, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner_dim, dim), nn.Dropout(dropout))

    @device_guard
    def forward(self, x, context=None, kv_include_self=False):
        context = x if context is None else context
        if kv_include_self:
            context = torch.cat((x, context), dim=1)
        out = ops.cross_attention(self.to_q(x), self.to_kv(context), self.heads, self.scale)
        return self.to_out(out)


class Transformer(nn.Module):
    def __init__(self, dim, depth, heads, dim_head, mlp_dim, dropout=0.):
        super().__init__()
        self.layers = nn.ModuleList([])
        self.norm = nn.LayerNorm(dim)
        self._drops = None
        for _ in range(depth):
            self.layers.append(nn.ModuleList([
                PreNorm(dim, Attention(dim, heads=heads, dim_head=dim_head, dropout=dropout)),
                PreNorm(dim, FeedForward(dim, mlp_dim, dropout=dropout))]))

    def _dropout_active(self):
        if self._drops is None:           # the module-tree walk costs ~0.5 ms: done once, the p values are read live
            self._drops = [m for m in self.modules() if isinstance(m, nn.Dropout) or hasattr(m, "tmf_keep_mask")]
        return any(m.training and (hasattr(m, "tmf_keep_mask") or m.p > 0) for m in self._drops)

    def _fused(self, x, allow_dropout=False):
        attn, ff = self.layers[0]
        a, f = attn.fn, ff.fn
        drop = (not allow_dropout) and self._dropout_active()
        return (not drop) and x.is_cuda and ops.fused_block_supported(x.shape[-1], a.to_q.out_features,
                                                                      f.net[0].out_features)

    @device_guard
    def forward(self, x, context=None, residual=None):
        """``residual`` (optional) is added to the result inside the final LayerNorm pass — the caller's
        ``enc(tokens, context) + tokens`` (reference networks.py:262-263) without a separate kernel."""
        # The fused layer reads keys / values from an UN-normalised context — what PreNorm hands the attention when
        # a context is passed (networks.py:120-121).  Without one the reference attends over LayerNorm(x) of the
        # CURRENT layer (networks.py:162 `default(context, x)` sees the normalised x), which the op-per-launch
        # branch below reproduces.
        if context is not None and self._fused(x):
            for attn, ff in self.layers:
                x = ops.transformer_layer(x, context, attn.norm, attn.fn, ff.norm, ff.fn)
        else:
            for attn, ff in self.layers:
                x = attn(x, context=context) + x
                x = ff(x) + x
        return ops.layer_norm(x, self.norm.weight, self.norm.bias, self.norm.eps, residual)


class _TokenReduce(nn.Module):
    """'b n d -> b d' by mean / max over the tokens: the reference's Rearrange + AdaptiveAvg/MaxPool1d(1) + Rearrange
    (networks.py:264-269) for callers that use ``fuse_transformer.gap`` / ``.gmp`` directly."""

    def __init__(self, how):
        super().__init__()
        self.how = how

    def forward(self, tokens):
        return tokens.mean(dim=1) if self.how == "mean" else tokens.amax(dim=1)


class CrossTransformer_MOD_AVG(nn.Module):
    """``depth`` x [mri <- Transformer(mri | pet) + mri ; pet <- Transformer(pet | NEW mri) + pet], then
    cat[mean, mean, max, max] over tokens -> (B, 4*dim).   reference: networks.py:255-281"""

    def __init__(self, dim, depth, heads, dim_head, mlp_dim, dropout):
        super().__init__()
        self.layers = nn.ModuleList([])
        for _ in range(depth):
            self.layers.append(nn.ModuleList([Transformer(dim, 1, heads, dim_head, mlp_dim, dropout=dropout),
                                              Transformer(dim, 1, heads, dim_head, mlp_dim, dropout=dropout)]))
        # attribute parity with networks.py:264-269 (parameter-free; forward() pools all four in ONE launch instead)
        self.gap = _TokenReduce("mean")
        self.gmp = _TokenReduce("max")

    def _one_call_ok(self, mri_tokens):
        """dim-128 geometry, depth-1 Transformer instances, dropout inactive, nobody listening on the inner modules:
        the whole fusion is one library call per pass (ops.FusionTrain)."""
        if not (mri_tokens.is_cuda and torch.is_grad_enabled()) or len(self.layers) == 0:
            return False
        t0 = self.layers[0][0]
        a, f = t0.layers[0][0].fn, t0.layers[0][1].fn
        inner = a.to_q.out_features
        if not ops.fusion_one_call_supported(mri_tokens.shape[-1], inner, f.net[0].out_features, inner // a.heads,
                                             len(self.layers)):
            return False
        # Dropout (options/option.py:39) stays on the HIP path where the fused per-instance kernels take the masks
        masks_ok = ops.fusion_fused_supported(mri_tokens.shape[1], mri_tokens.shape[-1], a.heads, inner // a.heads,
                                              f.net[0].out_features)
        for pair in self.layers:
            for tr in pair:
                if (len(tr.layers) != 1 or not tr._fused(mri_tokens, allow_dropout=masks_ok) or tr._forward_hooks
                        or tr._forward_pre_hooks):
                    return False
                at = tr.layers[0][0].fn
                if at.heads != a.heads or at.to_q.out_features != inner or at.scale != (inner // a.heads) ** -0.5:
                    return False
        return True

    @device_guard
    def forward(self, mri_tokens, pet_tokens):
        if self._one_call_ok(mri_tokens):
            params, eps, drops = [], [], []
            for pair in self.layers:
                for tr in pair:
                    pa, pf = tr.layers[0]
                    a, f = pa.fn, pf.fn
                    drops.append((a.to_out[1], f.net[2], f.net[4]))
                    params += [pa.norm.weight, pa.norm.bias, a.to_q.weight, a.to_kv.weight, a.to_out[0].weight,
                               a.to_out[0].bias, pf.norm.weight, pf.norm.bias, f.net[0].weight, f.net[0].bias,
                               f.net[3].weight, f.net[3].bias, tr.norm.weight, tr.norm.bias]
                    eps.append((float(pa.norm.eps), float(pf.norm.eps), float(tr.norm.eps)))
            a0 = self.layers[0][0].layers[0][0].fn
            f0 = self.layers[0][0].layers[0][1].fn
            cfg = (a0.heads, a0.to_q.out_features // a0.heads, f0.net[0].out_features, len(self.layers), tuple(eps),
                   tuple(drops) if any(tr._dropout_active() for pair in self.layers for tr in pair) else None)
            return ops.FusionTrain.apply(mri_tokens, pet_tokens, cfg, *params)
        for mri_enc, pet_enc in self.layers:
            # (Transformer.forward can fold this "+ tokens" into its last LayerNorm pass via residual=; it is left
            #  as its own add so that forward hooks on the Transformer modules see the reference's values)
            mri_tokens = mri_enc(mri_tokens, context=pet_tokens) + mri_tokens
            pet_tokens = pet_enc(pet_tokens, context=mri_tokens) + pet_tokens
        return ops.token_pool(mri_tokens, pet_tokens)
