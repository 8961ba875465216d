"""Regenerable inputs and parameters for the golden fixtures (TEST INFRASTRUCTURE).

Nothing here depends on the reference or on torch's RNG: every array comes from
numpy's frozen legacy MT19937 stream (``np.random.RandomState``), so the 16.7 MB
of weights and the volumes never need to be stored — only the seeds do
(SURVEY.md §8c "Fixture recipe").
"""
from __future__ import annotations

import numpy as np


def _fan(shape):
    """(fan_in, fan_out) as torch.nn.init computes them."""
    if len(shape) < 2:
        return shape[0], shape[0]
    rf = int(np.prod(shape[2:])) if len(shape) > 2 else 1
    return shape[1] * rf, shape[0] * rf


def init_arrays(spec, seed: int = 7) -> dict:
    """Per-tensor fill in ``spec`` (state_dict) order.

    conv weights   ~ N(0, sqrt(2/fan_out))   (same scale as kaiming_normal_(fan_out, relu),
                                              mymodel.py:195-199)
    other weights  ~ N(0, 1/sqrt(fan_in))
    conv / linear biases ~ N(0, 0.1)
    norm gammas    = 1 + 0.1 n,  norm betas = 0.1 n         (n ~ N(0,1))
    running_mean   = 0.1 n,  running_var = 1 + 0.1 |n|,  num_batches_tracked = 0
    """
    rs = np.random.RandomState(seed)
    out = {}
    for name, (kind, shape) in spec.items():
        if name.endswith("num_batches_tracked"):
            out[name] = np.zeros((), np.int64)
            continue
        n = rs.standard_normal(shape).astype(np.float64)
        is_norm = len(shape) == 1 and (".norm." in name or _is_bn_key(name, spec))
        if name.endswith("running_mean"):
            a = 0.1 * n
        elif name.endswith("running_var"):
            a = 1.0 + 0.1 * np.abs(n)
        elif is_norm:
            a = 1.0 + 0.1 * n if name.endswith("weight") else 0.1 * n
        elif len(shape) == 5:
            a = n * np.sqrt(2.0 / _fan(shape)[1])
        elif len(shape) == 2:
            a = n / np.sqrt(_fan(shape)[0])
        else:                       # biases
            a = 0.1 * n
        out[name] = a.astype(np.float32)
    return out


def _is_bn_key(name: str, spec) -> bool:
    """A 1-d '.weight'/'.bias' belongs to a BatchNorm iff its prefix also owns a running_mean."""
    prefix = name.rsplit(".", 1)[0]
    return (prefix + ".running_mean") in spec


def make_inputs(batch: int, size, seed: int = 1234, kind: str = "uniform"):
    """MRI first, then PET, from one stream, uniform [0,1) (the range ScaleIntensityd
    produces, datasets/ADNI.py:64); labels arange(B) % 2.  kind="blobs": structured volumes
    (make_inputs_blobs) whose pooled features differ from sample to sample."""
    if kind == "blobs":
        return make_inputs_blobs(batch, size, seed)
    if kind != "uniform":
        raise ValueError(kind)
    rs = np.random.RandomState(seed)
    shape = (batch, 1) + tuple(size)
    mri = rs.rand(*shape).astype(np.float32)
    pet = rs.rand(*shape).astype(np.float32)
    y = (np.arange(batch) % 2).astype(np.int64)
    return mri, pet, y


def make_inputs_blobs(batch: int, size, seed: int = 1234):
    """Structured volumes: per sample a noise floor + a linear ramp + four Gaussian blobs with sample-specific
    centres / widths / amplitudes, min-max scaled to [0, 1] per volume (what ScaleIntensityd does,
    datasets/ADNI.py:64).  Unlike uniform noise the pooled features of two samples differ by O(0.1), so the
    train-mode BatchNorm1d layers of the heads (batch of 2!) are well conditioned.  MRI volumes first, then PET."""
    rs = np.random.RandomState(seed)
    D, H, W = size
    zz = np.linspace(-1.0, 1.0, D, dtype=np.float32)[:, None, None]
    yy = np.linspace(-1.0, 1.0, H, dtype=np.float32)[None, :, None]
    xx = np.linspace(-1.0, 1.0, W, dtype=np.float32)[None, None, :]
    vols = []
    for _modality in range(2):
        out = np.empty((batch, 1, D, H, W), np.float32)
        for b in range(batch):
            v = (0.15 * rs.rand(D, H, W)).astype(np.float32)
            g = rs.uniform(-1.0, 1.0, 3).astype(np.float32)
            v += 0.2 * (g[0] * zz + g[1] * yy + g[2] * xx)
            for _ in range(4):
                c = rs.uniform(-0.6, 0.6, 3).astype(np.float32)
                sig = np.float32(rs.uniform(0.15, 0.45))
                amp = np.float32(rs.uniform(0.3, 1.0))
                r2 = (zz - c[0]) ** 2 + (yy - c[1]) ** 2 + (xx - c[2]) ** 2
                v += amp * np.exp(-r2 / (2.0 * sig * sig)).astype(np.float32)
            lo, hi = v.min(), v.max()
            out[b, 0] = (v - lo) / (hi - lo)
        vols.append(out)
    y = (np.arange(batch) % 2).astype(np.int64)
    return vols[0], vols[1], y


def make_masks(batch: int, seed: int = 99):
    """Keep-masks for the two Dropout(0.5) layers of model_ad.fc_cls (mymodel.py:190-191)."""
    rs = np.random.RandomState(seed)
    return rs.rand(batch, 512) >= 0.5, rs.rand(batch, 64) >= 0.5


def make_fusion_masks(rows: int, n_inst: int, p: float, dim: int = 128, mlp: int = 512, seed: int = 123):
    """Keep-masks (booleans) for the three Dropout(p) modules of every Transformer instance of the fusion block
    (networks.py:153 after to_out, :131 after GELU, :133 after the second Linear), instance order = execution order
    (mri encoder of layer 0, pet encoder of layer 0, mri encoder of layer 1, ...): [(rows x dim, rows x mlp, rows x dim)]."""
    rs = np.random.RandomState(seed)
    return [(rs.rand(rows, dim) >= p, rs.rand(rows, mlp) >= p, rs.rand(rows, dim) >= p) for _ in range(n_inst)]


def probe_indices(numel: int, k: int = 16, seed: int = 5):
    """Fixed flat indices at which tensors are sampled into the fixtures."""
    rs = np.random.RandomState(seed + (numel % 9973))
    return rs.randint(0, numel, size=min(k, numel)).astype(np.int64)
