#!/usr/bin/env python3
"""Where the host-fed input path spends its time (each phase of DevicePrefetcher._launch, timed on the host)."""
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
pf = T.DevicePrefetcher(pool, device=dev, flip_prob=0.3, seed=0)
T.scale_intensity_flip(torch.zeros((B, 1) + vol, device=dev))
torch.cuda.synchronize()
acc = {}


def tick(name, t0):
    torch.cuda.synchronize()
    if it >= 4:                    # steady state only (the first rounds allocate the pinned sets)
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0


for it in range(10):
    hb = pool[it % 3]
    slot = it & 1
    with torch.cuda.stream(pf.copy_stream):
        t0 = time.perf_counter(); a = torch.as_tensor(hb["MRI"]); tick("as_tensor", t0)
        t0 = time.perf_counter(); buf = pf._stage(slot, "MRI", hb["MRI"]); tick("stage (pageable -> pinned)", t0)
        t0 = time.perf_counter(); raw = buf.to(dev, non_blocking=True); tick("to(device)", t0)
        t0 = time.perf_counter(); out = T.scale_intensity_flip(raw, None, stream=pf.copy_stream); tick("scale kernels", t0)
    t0 = time.perf_counter(); fl = (pf.rs.random_sample(B) < 0.3).astype(np.uint8); tick("flip decisions", t0)
    t0 = time.perf_counter(); lb = pf._stage(slot, "label", np.asarray(hb["label"], dtype=np.int64)); tick("stage label", t0)
    t0 = time.perf_counter(); out2, ev = pf._launch(slot, hb); tick("_launch (whole batch: 2 modalities + labels)", t0)
for k, v in acc.items():
    print(f"{k:48s} {v / 6 * 1e3:8.3f} ms")
it = iter(T.DevicePrefetcher(itertools.cycle(pool), device=dev, flip_prob=0.3, seed=0))
for _ in range(5):
    next(it)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    b = next(it)
torch.cuda.synchronize()
print(f"DevicePrefetcher alone: {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms per batch of {B} pairs")
it.close()
