#!/usr/bin/env python3
"""Where the host-fed input path spends its time: pageable -> pinned staging, H2D, device transform, and the whole
DevicePrefetcher loop with nothing else running."""
import itertools
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import transmf_ad_amd as T      # noqa: E402

B, vol = 8, (96, 96, 96)
rs = np.random.RandomState(0)
pool = [dict(MRI=(rs.rand(B, 1, *vol) * 4000).astype(np.float32), PET=(rs.rand(B, 1, *vol) * 9).astype(np.float32),
             label=np.arange(B) % 2) for _ in range(3)]
dev = torch.device("cuda:0")
pin = torch.empty((B, 1) + vol, dtype=torch.float32, pin_memory=True)
t = torch.as_tensor(pool[0]["MRI"])
for _ in range(3):
    pin.copy_(t)
t0 = time.perf_counter()
for _ in range(10):
    pin.copy_(t)
dt = (time.perf_counter() - t0) / 10
print(f"pageable -> pinned copy of {t.numel() * 4 / 1e6:.1f} MB: {dt * 1e3:.2f} ms ({t.numel() * 4 / dt / 1e9:.1f} GB/s)")
d = torch.empty_like(pin, device=dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    d.copy_(pin, non_blocking=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print(f"pinned -> device: {dt * 1e3:.2f} ms ({t.numel() * 4 / dt / 1e9:.1f} GB/s)")
for _ in range(3):
    T.scale_intensity_flip(d)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    T.scale_intensity_flip(d)
torch.cuda.synchronize()
print(f"device ScaleIntensity (+flip) of one modality batch: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms")
it = iter(T.DevicePrefetcher(itertools.cycle(pool), device=dev, flip_prob=0.3, seed=0))
for _ in range(5):
    next(it)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    b = next(it)
torch.cuda.synchronize()
print(f"DevicePrefetcher alone: {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms per batch of {B} pairs")
