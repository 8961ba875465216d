"""CPU restatement of the input-pipeline step ahead of the hot path (TEST INFRASTRUCTURE, not product code: only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import anything under oracle/).

reference call sites: datasets/ADNI.py:64 `ScaleIntensityd(keys=['MRI', 'PET'])`, :66 `RandFlipd(keys=[...], prob=0.3,
spatial_axis=0)`; the transforms themselves live in MONAI (requirements.txt pins `monai`; the package is NOT vendored in
/root/reference and is not installed in this image), so this file restates MONAI's PUBLISHED algorithm:

  * monai.transforms.ScaleIntensity(minv=0.0, maxv=1.0, factor=None, channel_wise=False).__call__ ->
    monai.transforms.utils.rescale_array(arr, minv, maxv):
        mina = arr.min(); maxa = arr.max()
        if mina == maxa: return arr * minv  (minv is not None)
        norm = (arr - mina) / (maxa - mina)
        return norm * (maxv - minv) + minv
    evaluated in float32 (EnsureChannelFirstd output of a float32 NIfTI); with minv = 0, maxv = 1 the last line is the
    identity in IEEE arithmetic (x * 1.0 + 0.0 == x for every finite x >= 0).
  * monai.transforms.Flip(spatial_axis=0) on a channel-first (C, D, H, W) array = reverse axis 1 (the first spatial
    axis); RandFlipd draws ONE decision per dictionary (`self.R.random() < prob`), so MRI and PET of a subject flip
    together.  The decision comes from MONAI's own RandomState and is an INPUT here.

  * monai.transforms.RandRotated(range_x=0.05, prob=0.3) (ADNI.py:67; keep_size=True, mode="bilinear",
    padding_mode="border", align_corners=False): angle x ~ U(-0.05, 0.05) about the FIRST spatial axis (y = z = 0), then
    Rotate: transform = T(+c) @ Rx(angle) @ T(-c) with c = (size - 1) / 2 and Rx = [[1,0,0],[0,cos,-sin],[0,sin,cos]] on
    index coordinates, applied in the PULL direction (monai.networks.layers.AffineTransform(normalized=False,
    reverse_indexing=True): out[o] = in[transform @ o], the scipy.ndimage.affine_transform convention) through
    F.affine_grid + F.grid_sample(mode="bilinear", padding_mode="border", align_corners=False).  Rx leaves the first
    axis alone, so the sampling point of out[d, h, w] is (d, s1, s2) with
        s1 = c1 + (cos * (h - c1) - sin * (w - c2)),   s2 = c2 + (sin * (h - c1) + cos * (w - c2)),
    clipped to [0, size - 1] (border), interpolated bilinearly in the (H, W) plane d.  `rotate_x` below evaluates exactly
    that in fp32 with the operation order written there; MONAI's own route goes through normalised coordinates, which
    perturbs the sampling point by ~1e-6 voxels (tests/test_input_pipeline.py checks `rotate_x` against that route,
    built from torch's affine_grid / grid_sample on the host, to 2e-5).
  * monai.transforms.RandZoomd(min_zoom=0.95, max_zoom=1, prob=0.3) (ADNI.py:68; mode="area", padding_mode="edge",
    keep_size=True): ONE factor z ~ U(0.95, 1) for all three axes, Zoom: F.interpolate(mode="area") to
    size_k = floor(S_k * z) (= adaptive average pooling: output o averages input [floor(o S / So), ceil((o + 1) S / So)),
    sum in d, h, w order, divided by the three window lengths in turn), then padded back to the original size with
    edge replication, (S_k - So_k) // 2 voxels in front.  `zoom_area` is BIT-identical to torch's CPU
    F.interpolate(mode="area") + F.pad(mode="replicate") (checked in tests/test_input_pipeline.py).

Parity status: UNPINNED against MONAI itself (library absent: no golden vector can be generated); pinned to the
formulas above, and — for the two interpolating transforms — to torch's own grid_sample / interpolate on the host.
Every random decision (flip?, rotate? + angle, zoom? + factor) is an INPUT: MONAI draws them from its own RandomState.
"""
from __future__ import annotations

import math

import numpy as np


def scale_intensity(vol: np.ndarray, minv: float = 0.0, maxv: float = 1.0) -> np.ndarray:
    """rescale_array on ONE volume (any shape), float32 in / float32 out."""
    arr = np.asarray(vol, dtype=np.float32)
    mina, maxa = arr.min(), arr.max()
    if mina == maxa:
        return arr * np.float32(minv)
    norm = (arr - mina) / (maxa - mina)
    return (norm * np.float32(maxv - minv) + np.float32(minv)).astype(np.float32)


def rand_flip(vol: np.ndarray, do_flip: bool, spatial_axis: int = 0) -> np.ndarray:
    """Flip(spatial_axis) of a channel-first volume (C, D, H, W) when the (externally drawn) decision says so."""
    return np.flip(vol, axis=spatial_axis + 1).copy() if do_flip else vol


def rotate_x(vol: np.ndarray, angle: float) -> np.ndarray:
    """Rotate(angle=(angle, 0, 0), keep_size=True, bilinear, border) of a channel-first volume (C, D, H, W): pull-direction
    rotation in every (H, W) plane about its centre.  fp32 throughout, one rounding per written operation."""
    f = np.float32
    v = np.asarray(vol, dtype=np.float32)
    C, D, H, W = v.shape
    cs, sn = f(math.cos(angle)), f(math.sin(angle))
    c1, c2 = f((H - 1) / 2), f((W - 1) / 2)
    o1 = (np.arange(H, dtype=np.float32) - c1)[:, None]
    o2 = (np.arange(W, dtype=np.float32) - c2)[None, :]
    s1 = c1 + (cs * o1 - sn * o2)
    s2 = c2 + (sn * o1 + cs * o2)
    s1 = np.minimum(np.maximum(s1, f(0)), f(H - 1))
    s2 = np.minimum(np.maximum(s2, f(0)), f(W - 1))
    f1, f2 = np.floor(s1), np.floor(s2)
    t1, t2 = s1 - f1, s2 - f2
    a, b = f(1) - t1, f(1) - t2
    i1, i2 = f1.astype(np.int64), f2.astype(np.int64)
    j1, j2 = np.minimum(i1 + 1, H - 1), np.minimum(i2 + 1, W - 1)
    v00, v01, v10, v11 = v[:, :, i1, i2], v[:, :, i1, j2], v[:, :, j1, i2], v[:, :, j1, j2]
    out = ((v00 * (a * b) + v01 * (a * t2)) + v10 * (t1 * b)) + v11 * (t1 * t2)
    return out.astype(np.float32)


def zoom_out_size(shape, zoom: float):
    """Zoom's output size per axis: int(floor(float(S) * zoom))."""
    return tuple(int(math.floor(float(s) * zoom)) for s in shape)


def zoom_area(vol: np.ndarray, zoom: float) -> np.ndarray:
    """Zoom(zoom, mode="area", padding_mode="edge", keep_size=True) for zoom <= 1 on (C, D, H, W): adaptive average to
    floor(S * zoom) per axis, then edge padding back to the original size."""
    f = np.float32
    v = np.asarray(vol, dtype=np.float32)
    C, D, H, W = v.shape
    So = zoom_out_size((D, H, W), zoom)
    if any(o < 1 for o in So) or any(o > s for o, s in zip(So, (D, H, W))):
        raise ValueError(f"zoom {zoom} on {(D, H, W)}: only 0 < zoom <= 1 is restated (RandZoomd(0.95, 1))")
    ax = []
    for S, O in zip((D, H, W), So):
        o = np.arange(O, dtype=np.int64)
        start = (o * S) // O
        end = ((o + 1) * S + O - 1) // O
        ax.append((start, end - start))
    kmax = [int(k.max()) for _s, k in ax]
    (sd, kd), (sh, kh), (sw, kw) = ax
    acc = np.zeros((C,) + So, dtype=np.float32)
    for a in range(kmax[0]):
        for b in range(kmax[1]):
            for c in range(kmax[2]):
                valid = (a < kd)[:, None, None] & (b < kh)[None, :, None] & (c < kw)[None, None, :]
                idd = np.minimum(sd + a, D - 1)[:, None, None]
                ihh = np.minimum(sh + b, H - 1)[None, :, None]
                iww = np.minimum(sw + c, W - 1)[None, None, :]
                term = v[:, idd, ihh, iww]
                acc = np.where(valid[None], acc + term, acc)
    z = ((acc / kd.astype(np.float32)[None, :, None, None]) / kh.astype(np.float32)[None, None, :, None]) \
        / kw.astype(np.float32)[None, None, None, :]
    z = z.astype(np.float32)
    idx = [np.clip(np.arange(S) - (S - O) // 2, 0, O - 1) for S, O in zip((D, H, W), So)]
    return z[:, idx[0][:, None, None], idx[1][None, :, None], idx[2][None, None, :]].astype(np.float32)


def train_transform(batch_mri: np.ndarray, batch_pet: np.ndarray, flips, angles=None, zooms=None) -> tuple:
    """The deterministic part of ADNI_transform('True') (ADNI.py:59-70) on a collated batch (B, 1, D, H, W):
    per subject and modality ScaleIntensity, then — with the decisions shared by the two modalities of a subject — the
    flip flips[b], the rotation by angles[b] (None / NaN = not applied) and the zoom by zooms[b] (None / NaN = not applied)."""
    out = []
    for batch in (batch_mri, batch_pet):
        res = np.empty_like(batch, dtype=np.float32)
        for b in range(batch.shape[0]):
            x = rand_flip(scale_intensity(batch[b]), bool(flips[b]))
            if angles is not None and angles[b] is not None and not np.isnan(angles[b]):
                x = rotate_x(x, float(angles[b]))
            if zooms is not None and zooms[b] is not None and not np.isnan(zooms[b]):
                x = zoom_area(x, float(zooms[b]))
            res[b] = x
        out.append(res)
    return out[0], out[1]
