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

Parity status: UNPINNED against MONAI itself (library absent: no golden vector can be generated); pinned only to the
formulas above.  RandRotated / RandZoomd (ADNI.py:67-68) are not restated: they interpolate with MONAI-specific
conventions that cannot be checked here.
"""
from __future__ import annotations

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


def train_transform(batch_mri: np.ndarray, batch_pet: np.ndarray, flips) -> tuple:
    """The deterministic part of ADNI_transform('True') (ADNI.py:59-70) on a collated batch (B, 1, D, H, W):
    per subject and modality ScaleIntensity, then the shared flip decision flips[b]."""
    out = []
    for batch in (batch_mri, batch_pet):
        res = np.empty_like(batch, dtype=np.float32)
        for b in range(batch.shape[0]):
            res[b] = rand_flip(scale_intensity(batch[b]), bool(flips[b]))
        out.append(res)
    return out[0], out[1]
