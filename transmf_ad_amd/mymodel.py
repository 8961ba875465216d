"""Drop-in model classes (reference: models/mymodel.py): same constructor arguments, forward
signatures, return tuples, attribute names and state_dict keys as the reference's ``model_ad``
(:182-222), ``model_CNN_ad`` (:144-179) and ``model_single`` (:13-37), computed by the MI355X
HIP kernels.  ``kfold_train_adversarial.py``'s train_step/val_step run unchanged on them.

The small dense heads (``D``, ``fc_cls``, ``fc``: (B, <=512) matrices) stay on stock torch ops.
"""
from __future__ import annotations

import torch
from torch import nn

from .gradient_reversal import revgrad
from .networks import CrossTransformer_MOD_AVG, device_guard, sNet


def _init_like_reference(module: nn.Module) -> None:
    """mymodel.py:195-202: kaiming-normal(fan_out, relu) conv weights, BN3d gamma=1, beta=0."""
    for m in module.modules():
        if isinstance(m, nn.Conv3d):
            nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
        elif isinstance(m, nn.BatchNorm3d):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)


def _two_streams(mri_fn, mri, pet_fn, pet):
    """Run the two independent sNet encoders (mymodel.py:206-207) on two HIP streams of the same GPU.

    The big convolutions of either encoder fill the chip on their own; what overlaps is everything that does
    not: the deep 12^3 / 24^3 layers (1.7 workgroup rounds each), the finalize/reduce kernels, weight repacking.
    Autograd replays each encoder's backward on the stream its forward ran on, so backward overlaps too.
    TMF_STREAMS=1 disables it."""
    import os
    if os.environ.get("TMF_STREAMS", "2") == "1" or not mri.is_cuda:
        return mri_fn(mri), pet_fn(pet)
    cur = torch.cuda.current_stream(mri.device)
    side = _side_stream(mri.device)
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        pet_out = pet_fn(pet)
    mri_out = mri_fn(mri)
    cur.wait_stream(side)
    pet_out.record_stream(cur)
    return mri_out, pet_out


_SIDE = {}


def _side_stream(device):
    key = (device.type, device.index)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]


def backward_streams(device):
    """The streams, besides the caller's, that a backward of these models runs on (the side stream of _two_streams):
    what a data-parallel wrapper has to wait for before it reads gradients (parallel.GradAllReduce)."""
    s = _SIDE.get((device.type, device.index))
    return [] if s is None else [s]


def _tokens(emb):
    """'b d x y z -> b (x y z) d' (mymodel.py:218).  sNet returns a (B, C, d, h, w) VIEW of its
    channels-last buffer, so this is a reshape of strides only — no transpose kernel, no copy."""
    return emb.flatten(2).transpose(1, 2)


class _FastModeSwitch:
    """``.train()`` / ``.eval()`` as one flat loop over a cached module list.  The reference's train_step calls
    ``net_model.train()`` on every iteration (kfold_train_adversarial.py:104); nn.Module.train() re-walks the ~250
    sub-modules recursively (≈2.5 ms of host time — a quarter of a bf16-mode step here).  The cached list is checked
    against the live tree on every call (every child of every cached module must be cached, and nothing else) and
    rebuilt when a sub-module was added, removed or replaced anywhere below."""

    def set_precision(self, conv="fp32", storage="fp32"):
        """Convolution precision / activation storage of THIS model's encoders (sNet.set_precision), independent of other
        models in the process; conv=None returns to the process default."""
        for m in self.modules():
            if isinstance(m, sNet):
                m.set_precision(conv, storage)
        return self

    def tmf_backward_streams(self, device):
        """parallel.GradAllReduce asks the wrapped module which streams (besides the caller's) its backward uses."""
        return backward_streams(device)

    def _flat(self):
        cache = self.__dict__.get("_flat_modules")
        if cache is not None:
            flat, ids = cache
            n = 0
            ok = True
            for m in flat:
                for c in m._modules.values():
                    if c is not None:
                        n += 1
                        if id(c) not in ids:
                            ok = False
            if ok and n == len(flat) - 1:
                return flat
        flat = list(self.modules())
        self.__dict__["_flat_modules"] = (flat, {id(m) for m in flat})
        return flat

    def train(self, mode: bool = True):
        if not isinstance(mode, bool):
            raise ValueError("training mode is expected to be boolean")
        for m in self._flat():
            object.__setattr__(m, "training", mode)
        return self


class _Flatten5(nn.Module):
    """'b c x y z -> b (c x y z)' (stands in for the einops layer of the reference's ``gap``)."""

    def forward(self, x):
        return x.flatten(1)


def _discriminator(dim):
    return nn.Sequential(nn.Linear(dim, 128), nn.BatchNorm1d(128), nn.ReLU(), nn.Linear(128, 2))


def _plain(mods):
    """Nobody hooked into these modules (a hook would not see the one-launch path's intermediate tensors)."""
    return not any(m._forward_hooks or m._forward_pre_hooks or m._backward_hooks for m in mods)


def _cnn_heads_one_call_ok(model, fc, D, tok, M):
    """The heads of the CNN-only models as the reference builds them (mymodel.py:20, :148-151): fc = Linear-ReLU-Linear on the
    M concatenated token means, D = Linear-BatchNorm1d-ReLU-Linear or None — then they are one launch per direction
    (ops.HeadsCNN); anything else takes the module path."""
    from . import ops
    if not (ops.HEADS_ONE_CALL and tok.is_cuda and tok.dtype == torch.float32 and tok.shape[0] <= 32):
        return False
    dim = tok.shape[-1]
    if not (isinstance(fc, nn.Sequential) and [type(m) for m in fc] == [nn.Linear, nn.ReLU, nn.Linear]):
        return False
    if fc[0].in_features != M * dim or fc[2].in_features != fc[0].out_features or fc[0].bias is None or fc[2].bias is None:
        return False
    if dim % 16 or M * dim > 512 or fc[2].out_features > 16:          # csrc/heads.hip check_cnn_heads
        return False
    if D is not None and model.training and tok.shape[0] < 2:
        return False        # train-mode BatchNorm1d over one sample: the module path raises as the reference's does
    mods = list(fc) + [fc]
    if D is not None:
        if not (isinstance(D, nn.Sequential) and [type(m) for m in D] == [nn.Linear, nn.BatchNorm1d, nn.ReLU, nn.Linear]):
            return False
        if D[0].in_features != dim or D[3].in_features != D[0].out_features or D[3].out_features != fc[2].out_features \
                or D[0].bias is None or D[3].bias is None:
            return False
        bn = D[1]
        if not bn.affine or bn.momentum is None or (not bn.track_running_stats and not model.training):
            return False
        mods += list(D) + [D]
    if any(m.training != model.training for m in mods):               # one train / eval switch for the whole launch
        return False
    for m in mods:                                                    # what ops.HeadsCNN would refuse takes the module path
        for t in m.parameters(recurse=False):
            if not (t.is_cuda and t.device == tok.device and t.dtype == torch.float32 and t.is_contiguous()):
                return False
    return _plain(mods)


def _cnn_heads(model, fc, D, mri_tok, pet_tok):
    from . import ops
    params = [fc[0].weight, fc[0].bias, fc[2].weight, fc[2].bias]
    buffers, momentum, eps = (None, None), 0.0, 0.0
    if D is not None:
        bn = D[1]
        if model.training and bn.track_running_stats:
            bn.num_batches_tracked += 2                                # D runs twice
        if bn.track_running_stats:
            buffers = (bn.running_mean, bn.running_var)
        momentum, eps = float(bn.momentum), float(bn.eps)
        params += [D[0].weight, D[0].bias, bn.weight, bn.bias, D[3].weight, D[3].bias]
    return ops.HeadsCNN.apply(mri_tok, pet_tok, (model.training, momentum, eps, 2.0), buffers, *params)


class model_single(_FastModeSwitch, nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.cnn = sNet(dim)
        self.cnn.tmf_alone = True               # one encoder, one stream: its pooled layers take the register-tiled conv kernel
        self.avgpool = nn.AdaptiveAvgPool3d((1, 1, 1))
        self.fc = nn.Sequential(nn.Linear(128, 64), nn.ReLU(), nn.Linear(64, 2))
        _init_like_reference(self)

    @device_guard
    def forward(self, img):
        tok = _tokens(self.cnn(img))                            # (B, V, dim); avgpool + rearrange == mean over V
        if _cnn_heads_one_call_ok(self, self.fc, None, tok, 1):
            return _cnn_heads(self, self.fc, None, tok, None)
        return self.fc(tok.mean(dim=1))


class model_CNN_ad(_FastModeSwitch, nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.mri_cnn = sNet(dim)
        self.pet_cnn = sNet(dim)
        self.fc_cls = nn.Sequential(nn.Linear(dim * 2, 128), nn.ReLU(), nn.Linear(128, 2))
        self.gap = nn.Sequential(nn.AdaptiveAvgPool3d(1), _Flatten5())
        self.D = _discriminator(dim)
        _init_like_reference(self)

    @device_guard
    def forward(self, mri, pet):
        mri_emb, pet_emb = _two_streams(self.mri_cnn, mri, self.pet_cnn, pet)
        mri_tok, pet_tok = _tokens(mri_emb), _tokens(pet_emb)
        if _cnn_heads_one_call_ok(self, self.fc_cls, self.D, mri_tok, 2):
            return _cnn_heads(self, self.fc_cls, self.D, mri_tok, pet_tok)
        mri_feat = mri_tok.mean(dim=1)                          # == AdaptiveAvgPool3d(1) + flatten
        pet_feat = pet_tok.mean(dim=1)
        D_MRI_logits = self.D(revgrad(mri_feat, 2.0))
        D_PET_logits = self.D(revgrad(pet_feat, 2.0))
        output_logits = self.fc_cls(torch.cat([mri_feat, pet_feat], dim=1))
        return output_logits, D_MRI_logits, D_PET_logits


class model_ad(_FastModeSwitch, nn.Module):
    def __init__(self, dim, depth, heads, dim_head, mlp_dim, dropout):
        super().__init__()
        self.mri_cnn = sNet(dim)
        self.pet_cnn = sNet(dim)
        self.fuse_transformer = CrossTransformer_MOD_AVG(dim, depth, heads, dim_head, mlp_dim, dropout)
        self.fc_cls = nn.Sequential(nn.Linear(dim * 4, 512), nn.BatchNorm1d(512), nn.ReLU(), nn.Dropout(0.5),
                                    nn.Linear(512, 64), nn.BatchNorm1d(64), nn.ReLU(), nn.Dropout(0.5),
                                    nn.Linear(64, 2))
        self.gap = nn.Sequential(nn.AdaptiveAvgPool3d(1), _Flatten5())
        self.D = _discriminator(dim)
        _init_like_reference(self)

    @device_guard
    def forward_features(self, mri, pet):
        """-> (cls (B, 4*dim), D_MRI_logits, D_PET_logits); everything ahead of fc_cls."""
        mri_emb, pet_emb = _two_streams(self.mri_cnn, mri, self.pet_cnn, pet)
        mri_tok, pet_tok = _tokens(mri_emb), _tokens(pet_emb)      # (B, V, dim)
        D_MRI_logits = self.D(revgrad(mri_tok.mean(dim=1), 2.0))
        D_PET_logits = self.D(revgrad(pet_tok.mean(dim=1), 2.0))
        return self.fuse_transformer(mri_tok, pet_tok), D_MRI_logits, D_PET_logits

    def _heads_one_call_ok(self, tok):
        """The heads as the reference builds them (mymodel.py:190-194), nobody hooked into them, a batch the kernel
        holds in registers: fc_cls and both D calls are one launch per direction (ops.HeadsAD)."""
        from . import ops
        if not (ops.HEADS_ONE_CALL and tok.is_cuda and tok.dtype == torch.float32 and tok.shape[0] <= 32):
            return False
        fc, D = self.fc_cls, self.D
        if not (isinstance(fc, nn.Sequential) and len(fc) == 9 and isinstance(D, nn.Sequential) and len(D) == 4):
            return False
        want = (nn.Linear, nn.BatchNorm1d, nn.ReLU, None, nn.Linear, nn.BatchNorm1d, nn.ReLU, None, nn.Linear)
        for m, w in zip(fc, want):
            if w is None:
                if not (type(m) is nn.Dropout or hasattr(m, "tmf_keep_mask")):
                    return False
            elif type(m) is not w:
                return False
        if [type(m) for m in D] != [nn.Linear, nn.BatchNorm1d, nn.ReLU, nn.Linear]:
            return False
        dim = tok.shape[-1]
        if fc[0].in_features != 4 * dim or D[0].in_features != dim or fc[4].in_features != fc[0].out_features \
                or fc[8].in_features != fc[4].out_features or D[3].in_features != D[0].out_features \
                or fc[8].out_features != D[3].out_features:
            return False
        # the library's own preconditions (csrc/heads.hip check_desc): other widths take the module path, they do not raise
        if dim % 8 or fc[0].out_features % 32 or fc[8].out_features > 16:
            return False
        # one train / eval switch for the whole launch: a sub-module flipped on its own (model.D.eval()) takes the module path
        for m in list(fc) + list(D) + [fc, D]:
            if m.training != self.training:
                return False
        for m in (fc[0], fc[4], fc[8], D[0], D[3]):
            if m.bias is None:
                return False
        for bn in (fc[1], fc[5], D[1]):
            if not bn.affine or bn.momentum is None or (not bn.track_running_stats and not self.training):
                return False
        for m in list(fc) + list(D) + [fc, D]:
            if m._forward_hooks or m._forward_pre_hooks or m._backward_hooks:
                return False
        return True

    def _heads(self, cls, mri_tok, pet_tok):
        from . import ops
        fc, D = self.fc_cls, self.D
        B = cls.shape[0]
        # both keep-masks in ONE launch (ops.dropout_keep_masks -> tmf_dropout_keep_masks; a test stand-in with a fixed mask
        # supplies its own) instead of full + bernoulli + divide per mask
        masks = ops.dropout_keep_masks([(fc[3], (B, fc[0].out_features)), (fc[7], (B, fc[4].out_features))], cls.device)
        bns = (fc[1], fc[5], D[1])
        if self.training:
            nbt = [bn.num_batches_tracked for bn in bns if bn.track_running_stats]
            if nbt:
                torch._foreach_add_(nbt, [s for bn, s in zip(bns, (1, 1, 2)) if bn.track_running_stats])   # D runs twice
        buffers = []
        for bn in bns:
            buffers += [bn.running_mean, bn.running_var] if bn.track_running_stats else [None, None]
        cfg = (self.training, tuple(float(bn.momentum) for bn in bns), tuple(float(bn.eps) for bn in bns), 2.0)
        params = (fc[0].weight, fc[0].bias, fc[1].weight, fc[1].bias, fc[4].weight, fc[4].bias, fc[5].weight, fc[5].bias,
                  fc[8].weight, fc[8].bias, D[0].weight, D[0].bias, D[1].weight, D[1].bias, D[3].weight, D[3].bias)
        return ops.HeadsAD.apply(cls, mri_tok, pet_tok, masks[0], masks[1], cfg, buffers, *params)

    @device_guard
    def forward(self, mri, pet):
        mri_emb, pet_emb = _two_streams(self.mri_cnn, mri, self.pet_cnn, pet)
        mri_tok, pet_tok = _tokens(mri_emb), _tokens(pet_emb)      # (B, V, dim)
        if self._heads_one_call_ok(mri_tok):
            # fc_cls and the two discriminator calls in one launch; the fusion transformer sees the same tokens
            return self._heads(self.fuse_transformer(mri_tok, pet_tok), mri_tok, pet_tok)
        D_MRI_logits = self.D(revgrad(mri_tok.mean(dim=1), 2.0))
        D_PET_logits = self.D(revgrad(pet_tok.mean(dim=1), 2.0))
        return self.fc_cls(self.fuse_transformer(mri_tok, pet_tok)), D_MRI_logits, D_PET_logits
