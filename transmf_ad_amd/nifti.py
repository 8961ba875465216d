"""Minimal NIfTI-1 reader for the input step ahead of the hot path (host side, no third-party dependency).

reference call site: datasets/ADNI.py:62 `LoadImaged(keys=['MRI', 'PET'])` on `<subject>.nii.gz` files (:42-43), which
MONAI hands to nibabel: the array comes back as (X, Y, Z) float32 in the file's own orientation (MONAI's NibabelReader
default `as_closest_canonical=False`), intensity-scaled by the header's scl_slope / scl_inter when the slope is neither 0
nor NaN (nibabel's rule), and `EnsureChannelFirstd` (:63) puts a channel axis in front.  nibabel is not installed in
this image, so this file follows the PUBLISHED NIfTI-1 layout (nifti1.h): a 348-byte header — sizeof_hdr (int32 = 348
in the file's byte order), dim[8] int16 at byte 40, datatype int16 at 70, bitpix int16 at 72, vox_offset float32 at 108,
scl_slope / scl_inter float32 at 112 / 116, magic "n+1\\0" at 344 (single-file form) — then the voxel data at
vox_offset, first index fastest.  `.nii.gz` is the same stream gzip-compressed.  "Parity unpinned" against nibabel
itself; the tests build files with `write_nifti` and by hand, both byte orders.

`nifti_batches` turns lists of MRI / PET paths and labels into the host batches `DevicePrefetcher` consumes — the reading
and decompression then run in the prefetcher's worker thread, next to the training thread.
"""
from __future__ import annotations

import gzip
import struct
from typing import Iterable, Iterator, Sequence

import numpy as np

# NIfTI-1 datatype codes (nifti1.h) -> numpy
_DTYPES = {2: "u1", 4: "i2", 8: "i4", 16: "f4", 64: "f8", 256: "i1", 512: "u2", 768: "u4", 1024: "i8", 1280: "u8"}
_CODES = {np.dtype(v).str[1:]: k for k, v in _DTYPES.items()}


class NiftiError(ValueError):
    pass


def _read_all(path: str) -> bytes:
    with open(path, "rb") as f:
        head = f.read(2)
        f.seek(0)
        if head == b"\x1f\x8b":                      # gzip magic, whatever the file is called
            with gzip.GzipFile(fileobj=f) as g:
                return g.read()
        return f.read()


def read_nifti(path: str, dtype=np.float32) -> np.ndarray:
    """-> (X, Y, Z[, T...]) array of `dtype`, scaled by scl_slope / scl_inter as nibabel does."""
    raw = _read_all(path)
    if len(raw) < 348:
        raise NiftiError(f"{path}: {len(raw)} bytes is shorter than a NIfTI-1 header")
    for bo in ("<", ">"):
        if struct.unpack_from(bo + "i", raw, 0)[0] == 348:
            break
    else:
        raise NiftiError(f"{path}: sizeof_hdr is not 348 in either byte order (not NIfTI-1)")
    magic = raw[344:348]
    if magic not in (b"n+1\x00", b"ni1\x00"):
        raise NiftiError(f"{path}: bad magic {magic!r}")
    if magic == b"ni1\x00":
        raise NiftiError(f"{path}: two-file (.hdr / .img) NIfTI is not supported")
    dim = struct.unpack_from(bo + "8h", raw, 40)
    ndim = dim[0]
    if not 1 <= ndim <= 7:
        raise NiftiError(f"{path}: dim[0] = {ndim}")
    shape = tuple(int(d) for d in dim[1:1 + ndim])
    if any(d < 1 for d in shape):
        raise NiftiError(f"{path}: non-positive dimension in {shape}")
    code, bitpix = struct.unpack_from(bo + "2h", raw, 70)
    if code not in _DTYPES:
        raise NiftiError(f"{path}: datatype code {code} is not supported")
    dt = np.dtype(bo + _DTYPES[code])
    if dt.itemsize * 8 != bitpix:
        raise NiftiError(f"{path}: bitpix {bitpix} does not match datatype {code}")
    vox_offset, slope, inter = struct.unpack_from(bo + "3f", raw, 108)
    # nibabel's Nifti1Header.get_slope_inter: slope 0 or non-finite (NaN, inf) = "no scaling"; a valid slope with a
    # non-finite intercept is a header error
    scaled = slope != 0.0 and np.isfinite(slope)
    if scaled and not np.isfinite(inter):
        raise NiftiError(f"{path}: valid scl_slope {slope} but non-finite scl_inter {inter}")
    off = int(vox_offset) if vox_offset >= 352 else 352
    n = int(np.prod(shape))
    if len(raw) < off + n * dt.itemsize:
        raise NiftiError(f"{path}: {len(raw)} bytes, need {off + n * dt.itemsize} for shape {shape}")
    data = np.frombuffer(raw, dtype=dt, count=n, offset=off).reshape(shape, order="F")
    # (X, Y, Z): first index fastest in the file.  Scaling as nibabel's array proxy: in float64, then the caller's dtype
    if scaled and not (slope == 1.0 and inter == 0.0):
        out = (data.astype(np.float64) * np.float64(slope) + np.float64(inter)).astype(dtype)
    else:
        out = data.astype(dtype)
    return np.ascontiguousarray(out)


def write_nifti(path: str, arr: np.ndarray, slope: float = 0.0, inter: float = 0.0, big_endian: bool = False) -> None:
    """Write a single-file NIfTI-1 (`.nii`, or gzip-compressed when the name ends with `.gz`): identity affine, the
    array's dtype, optional scl_slope / scl_inter.  For tests and synthetic data sets."""
    a = np.asarray(arr)
    key = a.dtype.str[1:]
    if key not in _CODES:
        raise NiftiError(f"dtype {a.dtype} has no NIfTI-1 code")
    bo = ">" if big_endian else "<"
    hdr = bytearray(352)
    struct.pack_into(bo + "i", hdr, 0, 348)
    dims = [a.ndim] + list(a.shape) + [1] * (7 - a.ndim)
    struct.pack_into(bo + "8h", hdr, 40, *dims)
    struct.pack_into(bo + "2h", hdr, 70, _CODES[key], a.dtype.itemsize * 8)
    struct.pack_into(bo + "8f", hdr, 76, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0)     # pixdim
    struct.pack_into(bo + "3f", hdr, 108, 352.0, float(slope), float(inter))
    struct.pack_into(bo + "h", hdr, 254, 1)                                            # sform_code: scanner
    struct.pack_into(bo + "12f", hdr, 280, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0)         # srow_x / _y / _z
    hdr[344:348] = b"n+1\x00"
    payload = bytes(hdr) + a.astype(a.dtype.newbyteorder(bo)).tobytes(order="F")
    if path.endswith(".gz"):
        with gzip.open(path, "wb", compresslevel=1) as f:
            f.write(payload)
    else:
        with open(path, "wb") as f:
            f.write(payload)


def _volume_3d(path: str) -> np.ndarray:
    """One subject's (X, Y, Z) volume: trailing singleton axes of a 4-D+ file (dim[0] = 4 with dim[4] = 1, common for
    ADNI exports) are dropped — what MONAI's LoadImaged + EnsureChannelFirstd make of such a file is (1, X, Y, Z) — and
    anything that is not 3-D after that is an error rather than a volume whose axes the transforms would misread."""
    a = read_nifti(path)
    while a.ndim > 3 and a.shape[-1] == 1:
        a = a[..., 0]
    if a.ndim != 3:
        raise NiftiError(f"{path}: shape {a.shape} is not a 3-D volume")
    return a


def nifti_batches(mri_paths: Sequence[str], pet_paths: Sequence[str], labels: Sequence[int], batch_size: int,
                  drop_last: bool = False, order: Iterable[int] = None) -> Iterator[dict]:
    """Host batches {'MRI': (B, 1, X, Y, Z) float32, 'PET': ..., 'label': (B,) int64} read from NIfTI files — what
    DataLoader(ADNI(...), batch_size) yields ahead of the transforms (datasets/ADNI.py:42-46, datasets/__init__.py:56).
    Feed it to DevicePrefetcher: the files are then read in its worker thread while the previous batch trains."""
    if not (len(mri_paths) == len(pet_paths) == len(labels)):
        raise ValueError("mri_paths, pet_paths and labels must have the same length")
    idx = list(range(len(labels))) if order is None else list(order)
    for s in range(0, len(idx), batch_size):
        sel = idx[s:s + batch_size]
        if drop_last and len(sel) < batch_size:
            return
        mri = np.stack([_volume_3d(mri_paths[i])[None] for i in sel])
        pet = np.stack([_volume_3d(pet_paths[i])[None] for i in sel])
        yield {"MRI": mri, "PET": pet, "label": np.asarray([labels[i] for i in sel], dtype=np.int64)}
