#!/usr/bin/env python3
"""Host -> device copy rates on this box for the shapes the input pipeline moves (28.3 MB per modality batch)."""
import time
import numpy as np
import torch

dev = torch.device("cuda:0")
n = 8 * 96 * 96 * 96
src = np.random.RandomState(0).rand(n).astype(np.float32)
t_src = torch.from_numpy(src)
d = torch.empty(n, device=dev)


def timeit(fn, reps=8):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


pin = torch.empty(n, dtype=torch.float32, pin_memory=True)
print("is_pinned:", pin.is_pinned())
pin.copy_(t_src)
dt = timeit(lambda: d.copy_(pin, non_blocking=True))
print(f"pinned (written once) -> device      : {dt * 1e3:7.2f} ms  {n * 4 / dt / 1e9:6.1f} GB/s")
dt = timeit(lambda: pin.copy_(t_src))
print(f"pageable -> pinned (host memcpy)     : {dt * 1e3:7.2f} ms  {n * 4 / dt / 1e9:6.1f} GB/s")


def both():
    pin.copy_(t_src)
    d.copy_(pin, non_blocking=True)


dt = timeit(both)
print(f"host write + pinned -> device        : {dt * 1e3:7.2f} ms  {n * 4 / dt / 1e9:6.1f} GB/s")
dt = timeit(lambda: d.copy_(t_src))
print(f"pageable -> device (blocking)        : {dt * 1e3:7.2f} ms  {n * 4 / dt / 1e9:6.1f} GB/s")
pin5 = torch.empty((8, 1, 96, 96, 96), dtype=torch.float32, pin_memory=True)
d5 = torch.empty((8, 1, 96, 96, 96), device=dev)
pin5.copy_(t_src.view(8, 1, 96, 96, 96))
dt = timeit(lambda: d5.copy_(pin5, non_blocking=True))
print(f"pinned 5-D -> device                 : {dt * 1e3:7.2f} ms  {n * 4 / dt / 1e9:6.1f} GB/s")
dt = timeit(lambda: pin5.to(dev, non_blocking=True))
print(f"pinned 5-D .to(device) (fresh alloc) : {dt * 1e3:7.2f} ms  {n * 4 / dt / 1e9:6.1f} GB/s")
s = torch.cuda.Stream()
def side():
    with torch.cuda.stream(s):
        d5.copy_(pin5, non_blocking=True)
dt = timeit(side)
print(f"pinned 5-D -> device on a side stream: {dt * 1e3:7.2f} ms  {n * 4 / dt / 1e9:6.1f} GB/s")
