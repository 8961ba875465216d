"""Input pipeline ahead of the hot path, on the device.

The reference feeds `train_step` from a MONAI `DataLoader(num_workers=0)` (datasets/__init__.py:56): NIfTI load,
`ScaleIntensityd`, `RandFlipd(prob=0.3, spatial_axis=0)`, `RandRotated(prob=0.3, range_x=0.05)`,
`RandZoomd(prob=0.3, min_zoom=0.95, max_zoom=1)` on the host (datasets/ADNI.py:59-70), then `batch['MRI'].to(device)`
inside the step (kfold_train_adversarial.py:106-108).  Here the RAW volumes go to the device on a copy stream (the copy
of batch i + 1 overlaps the step of batch i) and all four transforms run there as HIP kernels
(csrc/input_pipeline.hip, bit-identical to MONAI's published algorithms as restated in oracle/input_oracle.py).  The
random DECISIONS (flip?, rotate? and the angle, zoom? and the factor) are drawn on the host with numpy, one set per
subject shared by MRI and PET as MONAI's dictionary transforms do (MONAI draws them from its own RandomState; only the
probabilities and ranges are part of the reference's configuration).  `nifti.read_nifti` / `nifti_batches` read
`.nii` / `.nii.gz` volumes in the prefetcher's worker thread (datasets/ADNI.py:62 LoadImaged).
"""
from __future__ import annotations

from typing import Iterable, Iterator, Optional

import numpy as np
import torch

from . import _lib


def scale_intensity_flip(vol: torch.Tensor, flips: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                         stream: Optional[torch.cuda.Stream] = None) -> torch.Tensor:
    """ScaleIntensity (per volume min-max to [0, 1]) and Flip(spatial_axis=0) of a device batch (B, 1, D, H, W) / (B, D, H, W).
    flips: uint8 device tensor (B,), non-zero = reverse the first spatial axis of that volume."""
    if not vol.is_cuda or vol.dtype != torch.float32:
        raise _lib.TmfError("scale_intensity_flip needs a float32 tensor on the HIP device (there is no CPU fallback)")
    v = vol.contiguous()
    B = v.shape[0]
    D, H, W = v.shape[-3:]
    if v.numel() != B * D * H * W:
        raise _lib.TmfError(f"expected (B, 1, D, H, W) or (B, D, H, W), got {tuple(vol.shape)}")
    if flips is not None and not (flips.is_cuda and flips.dtype == torch.uint8 and flips.numel() == B):
        raise _lib.TmfError("flips must be a uint8 device tensor of B elements")
    with torch.cuda.device(v.device), (torch.cuda.stream(stream) if stream is not None else _Null()):
        # allocations and launches under the SAME current stream (the caching allocator ties a block to its stream)
        out = torch.empty_like(v) if out is None else out
        s = torch.cuda.current_stream(v.device).cuda_stream
        nws = _lib.query("tmf_scale_intensity_workspace_bytes", B)
        ws = torch.empty(nws // 4 + 2 * B, device=v.device, dtype=torch.float32)
        minmax = ws[nws // 4:]
        _lib.call("tmf_volume_minmax", v.data_ptr(), minmax.data_ptr(), ws.data_ptr(), nws, B, D * H * W, s)
        _lib.call("tmf_scale_flip", v.data_ptr(), out.data_ptr(), minmax.data_ptr(),
                  None if flips is None else flips.data_ptr(), B, D, H, W, s)
    return out


def rotate_zoom(vol: torch.Tensor, angles=None, zooms=None, stream: Optional[torch.cuda.Stream] = None) -> torch.Tensor:
    """RandRotated / RandZoomd (datasets/ADNI.py:67-68) of a device batch (B, 1, D, H, W) with the decisions given: angles[b]
    (radians about the first spatial axis; NaN / None = no rotation), zooms[b] (factor in (0, 1]; NaN / None = no zoom).
    Bit-identical to oracle/input_oracle.py rotate_x / zoom_area."""
    import math
    if not vol.is_cuda or vol.dtype != torch.float32:
        raise _lib.TmfError("rotate_zoom needs a float32 tensor on the HIP device (there is no CPU fallback)")
    v = vol.contiguous()
    B = v.shape[0]
    D, H, W = v.shape[-3:]
    if v.numel() != B * D * H * W:
        raise _lib.TmfError(f"expected (B, 1, D, H, W) or (B, D, H, W), got {tuple(vol.shape)}")
    def decisions(vals):
        if vals is None:
            return [None] * B
        if len(vals) != B:
            raise _lib.TmfError(f"expected {B} per-volume decisions, got {len(vals)}")
        return [None if (x is None or math.isnan(float(x))) else float(x) for x in vals]
    ang, zs = decisions(angles), decisions(zooms)
    with torch.cuda.device(v.device), (torch.cuda.stream(stream) if stream is not None else _Null()):
        s = torch.cuda.current_stream(v.device).cuda_stream
        if any(a is not None for a in ang):
            cs = np.zeros((B, 2), np.float32)
            flag = np.zeros(B, np.uint8)
            for b_, a in enumerate(ang):
                if a is not None:
                    cs[b_] = (np.float32(math.cos(a)), np.float32(math.sin(a)))
                    flag[b_] = 1
            out = torch.empty_like(v)
            cs_d, flag_d = torch.from_numpy(cs).to(v.device), torch.from_numpy(flag).to(v.device)     # held until the call returns
            _lib.call("tmf_rotate_x", v.data_ptr(), out.data_ptr(), cs_d.data_ptr(), flag_d.data_ptr(), B, D, H, W, s)
            v = out
        if any(z is not None for z in zs):
            sz = np.zeros((B, 3), np.int32)
            flag = np.zeros(B, np.uint8)
            for b_, z in enumerate(zs):
                if z is not None:
                    o = [int(math.floor(float(n) * z)) for n in (D, H, W)]
                    if min(o) < 1 or any(a > n for a, n in zip(o, (D, H, W))):
                        raise _lib.TmfError(f"zoom factor {z}: only 0 < zoom <= 1 is provided (RandZoomd(0.95, 1))")
                    sz[b_] = o
                    flag[b_] = 1
            out = torch.empty_like(v)
            sz_d, flag_d = torch.from_numpy(sz).to(v.device), torch.from_numpy(flag).to(v.device)
            _lib.call("tmf_zoom_area", v.data_ptr(), out.data_ptr(), sz_d.data_ptr(), flag_d.data_ptr(), B, D, H, W, s)
            v = out
    return v.view(vol.shape)


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class DevicePrefetcher:
    """Iterate device batches {'MRI', 'PET', 'label'} from an iterable of HOST batches with the same keys (numpy arrays
    or CPU tensors; MRI / PET raw float32 volumes (B, 1, D, H, W)).  Batch i + 1 is staged into pinned memory and copied
    on a side stream while the caller trains on batch i; ScaleIntensity + flip run on that side stream too."""

    def __init__(self, batches: Iterable, device="cuda", flip_prob: float = 0.3, seed: Optional[int] = None,
                 train: bool = True, rotate_prob: float = 0.3, rotate_range: float = 0.05, zoom_prob: float = 0.3,
                 zoom_range=(0.95, 1.0), pinned_staging: bool = False):
        """train=True applies the reference's whole augmentation set with its probabilities and ranges
        (datasets/ADNI.py:66-68); set a probability to 0 to leave a transform out.  train=False: ScaleIntensity only
        (the reference's test_transform)."""
        self.batches = batches
        self.device = torch.device(device)
        self.flip_prob = flip_prob if train else 0.0
        self.rotate_prob = rotate_prob if train else 0.0
        self.rotate_range = float(rotate_range)
        self.zoom_prob = zoom_prob if train else 0.0
        self.zoom_range = (float(zoom_range[0]), float(zoom_range[1]))
        self.rs = np.random.RandomState(seed)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self._pinned = [dict(), dict()]          # two staging sets, reused while shapes stay the same
        # pinned_staging=False (default): the worker thread copies straight from the loader's pageable arrays with a
        # blocking .to(device) on the copy stream — the runtime stages through its own pinned pool at the full link rate
        # (50 GB/s measured here), and only the WORKER blocks.  Re-filling our own pinned set every batch measured 15 ms
        # per 28 MB on this box (tools/prefetch_probe.py), slower than the copy it was meant to speed up.
        self.pinned_staging = pinned_staging

    def _stage(self, slot, key, arr):
        t = torch.as_tensor(arr)
        if not self.pinned_staging:
            return t
        buf = self._pinned[slot].get(key)
        if buf is None or buf.shape != t.shape or buf.dtype != t.dtype:
            buf = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            self._pinned[slot][key] = buf
        buf.copy_(t)
        return buf

    def _launch(self, slot, host_batch):
        B = len(host_batch["label"])
        # one set of decisions per subject, shared by MRI and PET (MONAI dictionary transforms); per transform the "apply?"
        # draw first, then its parameters (RandRotated draws x, y, z — y and z from (0, 0); RandZoomd one factor)
        flips = (self.rs.random_sample(B) < self.flip_prob).astype(np.uint8)
        angles = np.full(B, np.nan)
        zooms = np.full(B, np.nan)
        if self.rotate_prob > 0:
            for b in range(B):
                if self.rs.random_sample() < self.rotate_prob:
                    angles[b] = self.rs.uniform(-self.rotate_range, self.rotate_range)
                    self.rs.uniform(0.0, 0.0); self.rs.uniform(0.0, 0.0)
        if self.zoom_prob > 0:
            for b in range(B):
                if self.rs.random_sample() < self.zoom_prob:
                    zooms[b] = self.rs.uniform(*self.zoom_range)
        with torch.cuda.stream(self.copy_stream):
            out = {}
            nb = self.pinned_staging
            fl = self._stage(slot, "_flips", flips).to(self.device, non_blocking=nb)
            for key in ("MRI", "PET"):
                raw = self._stage(slot, key, host_batch[key]).to(self.device, non_blocking=nb)
                x = scale_intensity_flip(raw, fl if self.flip_prob > 0 else None, stream=self.copy_stream)
                if not (np.isnan(angles).all() and np.isnan(zooms).all()):
                    x = rotate_zoom(x, angles, zooms, stream=self.copy_stream)
                out[key] = x
            out["label"] = self._stage(slot, "label", np.asarray(host_batch["label"], dtype=np.int64)).to(
                self.device, non_blocking=nb)
            out["_flips"], out["_angles"], out["_zooms"] = flips, angles, zooms
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        return out, ev

    def __iter__(self) -> Iterator[dict]:
        """A worker thread stages batch i + 1 (pageable -> pinned memcpy, H2D and the two kernels on the copy stream) while
        the caller trains on batch i: the main thread only waits on an event.  (Staging from the training thread itself
        serialises a 57 MB host memcpy per step with the kernel launches: 36 ms per step instead of 17.)"""
        import queue
        import threading
        q: "queue.Queue" = queue.Queue(maxsize=1)          # one batch ready + one being staged = the two pinned sets
        stop = threading.Event()
        dev = self.device

        def worker():
            try:
                torch.cuda.set_device(dev)
                events = [None, None]
                slot = 0
                for host_batch in self.batches:
                    if stop.is_set():
                        break
                    if events[slot] is not None:
                        events[slot].synchronize()          # the copies out of this pinned set have completed
                    out, ev = self._launch(slot, host_batch)
                    events[slot] = ev
                    slot ^= 1
                    while not stop.is_set():
                        try:
                            q.put((out, ev), timeout=0.1)
                            break
                        except queue.Full:
                            continue
                q.put(None)
            except BaseException as e:                       # surface worker failures in the training thread
                q.put(e)

        # The training thread is busy issuing launches for most of a step and only hands the GIL over every
        # sys.getswitchinterval() = 5 ms; the worker needs it ~30 times per batch (between its GIL-free copies and
        # launches), so with the default interval a 1 ms staging job stretches to tens of ms and the step waits for it
        # (measured: 36 ms per step instead of 17).  A 100 us interval while the prefetcher runs costs the trainer nothing
        # measurable and lets the worker through.
        import sys
        old_interval = sys.getswitchinterval()
        sys.setswitchinterval(1e-4)
        th = threading.Thread(target=worker, name="tmf-prefetch", daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    return
                if isinstance(item, BaseException):
                    raise item
                out, ev = item
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(ev)
                for k in ("MRI", "PET", "label"):
                    out[k].record_stream(cur)
                yield out
        finally:
            sys.setswitchinterval(old_interval)
            stop.set()
            while th.is_alive():
                try:
                    q.get_nowait()
                except queue.Empty:
                    pass
                th.join(timeout=0.05)
