"""Adam with the whole update in ONE launch (csrc/adam.hip, tmf_adam_step).

reference: utils/utils.py:38-39 (getOptimizer -> torch.optim.Adam(net.parameters(), lr=1e-4, betas default, weight decay 0)),
stepped once per train step at kfold_train_adversarial.py:135.  Same constructor arguments, update rule and skipping of
parameters without a gradient as torch.optim.Adam (amsgrad / maximize / capturable / foreach are not offered); the
moment estimates of a parameter group live in two flat buffers, `state_dict()` exposes them per parameter in torch's
layout (`step`, `exp_avg`, `exp_avg_sq` views) so a checkpoint loads into torch.optim.Adam and back.
"""
import ctypes as C
import math

import torch

from . import _lib


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError(f"invalid Adam hyper-parameters lr={lr} betas={betas} eps={eps} weight_decay={weight_decay}")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self._flat = {}                      # id(group) -> (exp_avg, exp_avg_sq, offsets, numel array, param pointer array)

    def __setstate__(self, state):
        """copy.deepcopy / pickle: torch serialises only defaults, state and param_groups — the kernel-side flat moment
        buffers and pointer tables are rebuilt from the per-parameter state (as after load_state_dict)."""
        super().__setstate__(state)
        self._rebuild_flat()

    def _group_state(self, group):
        if getattr(self, "_flat", None) is None:
            self._flat = {}
        st = self._flat.get(id(group))
        if st is not None:
            return st
        ps = group["params"]
        if not ps:
            return None
        dev = ps[0].device
        for p in ps:
            if p.dtype != torch.float32 or p.device != dev or not p.is_cuda or not p.is_contiguous():
                raise _lib.TmfError("transmf_ad_amd.optim.Adam: parameters must be contiguous float32 tensors on one HIP device")
        chunks = []
        for s in range(0, len(ps), _lib.ADAM_MAX_TENSORS):
            sub = ps[s:s + _lib.ADAM_MAX_TENSORS]
            numel = (C.c_long * len(sub))(*[p.numel() for p in sub])
            total = _lib.query("tmf_adam_state_elems", len(sub), numel)
            m = torch.zeros(total, device=dev, dtype=torch.float32)
            v = torch.zeros(total, device=dev, dtype=torch.float32)
            off, offs = 0, []
            for p in sub:
                offs.append(off)
                off += (p.numel() + 3) & ~3
            pptr = (C.c_void_p * len(sub))(*[p.data_ptr() for p in sub])
            # step counts as Python ints (no 156 x `.item()` per step): [0] = the count every parameter of the chunk shares,
            # [1] = {index: count} once some parameter sat out a step (torch's bias correction is per parameter), [2] = the
            # CPU tensor that every state[p]["step"] of the chunk IS while the counts agree (one add_ per step keeps torch's
            # state layout current); after a divergence each parameter gets a tensor of its own
            shared = torch.tensor(0.0)           # ONE `step` tensor for the whole chunk while the counts agree
            chunks.append((sub, m, v, offs, numel, pptr, [0, None, shared]))
            for p, o in zip(sub, offs):          # torch's per-parameter layout, as views of the flat buffers
                self.state[p] = {"step": shared, "exp_avg": m[o:o + p.numel()].view_as(p),
                                 "exp_avg_sq": v[o:o + p.numel()].view_as(p)}
        self._flat[id(group)] = chunks
        return chunks

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            chunks = self._group_state(group)
            if not chunks:
                continue
            b1, b2 = group["betas"]
            for sub, m, v, _offs, numel, pptr, cnt in chunks:
                grads = []
                missing = False
                for i, p in enumerate(sub):
                    if p.data_ptr() != pptr[i]:
                        pptr[i] = p.data_ptr()                      # parameter storage replaced (e.g. by .to())
                    g = p.grad
                    if g is None:
                        grads.append(None)
                        missing = True
                        continue
                    if g.is_sparse or g.dtype != torch.float32 or g.device != p.device:
                        raise _lib.TmfError("transmf_ad_amd.optim.Adam: gradients must be dense float32 on the parameter's device")
                    grads.append(g if g.is_contiguous() else g.contiguous())
                if not missing and cnt[1] is None:                   # every step of a normal run: ONE count for the chunk
                    cnt[0] += 1
                    cnt[2].add_(1.0)
                    steps = None
                    todo = (cnt[0],)
                else:                                                # some parameter sat out a step, now or earlier
                    if cnt[1] is None:
                        cnt[1] = {i: cnt[0] for i in range(len(sub))}
                        for p in sub:
                            self.state[p]["step"] = torch.tensor(float(cnt[0]))
                    for i, g in enumerate(grads):
                        if g is not None:
                            cnt[1][i] += 1
                            self.state[sub[i]]["step"] += 1
                    steps = [cnt[1][i] if g is not None else None for i, g in enumerate(grads)]
                    todo = sorted({s_ for s_ in steps if s_ is not None})
                # one launch per distinct step count (one, unless some parameter sat out earlier steps)
                for step in todo:
                    gptr = (C.c_void_p * len(sub))(*[g.data_ptr() if (g is not None and (steps is None or steps[i] == step))
                                                     else None for i, g in enumerate(grads)])
                    with torch.cuda.device(m.device):
                        _lib.call("tmf_adam_step", len(sub), pptr, gptr, numel, m.data_ptr(), v.data_ptr(), float(group["lr"]),
                                  float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]), step,
                                  torch.cuda.current_stream().cuda_stream)
        return loss

    def load_state_dict(self, state_dict):
        """torch's loader replaces the per-parameter state tensors: copy them back into the flat buffers the kernel uses."""
        super().load_state_dict(state_dict)
        self._rebuild_flat()

    def _rebuild_flat(self):
        self._flat = {}
        loaded = {p: dict(st) for p, st in self.state.items()}
        if not loaded:
            return
        for group in self.param_groups:
            for sub, m, v, offs, _numel, _pptr, cnt in self._group_state(group) or ():
                counts = []
                for p, o in zip(sub, offs):
                    old = loaded.get(p)
                    n = 0
                    if old and "exp_avg" in old:
                        m[o:o + p.numel()].copy_(old["exp_avg"].reshape(-1))
                        v[o:o + p.numel()].copy_(old["exp_avg_sq"].reshape(-1))
                        n = int(float(old.get("step", 0.0)))
                        self.state[p]["step"] = torch.tensor(float(n))
                    counts.append(n)
                if len(set(counts)) <= 1:
                    cnt[0], cnt[1] = (counts[0] if counts else 0), None
                    cnt[2].fill_(float(cnt[0]))
                    for p in sub:
                        self.state[p]["step"] = cnt[2]
                else:
                    cnt[0], cnt[1] = max(counts), dict(enumerate(counts))
                    # diverged counts: EVERY parameter gets a step tensor of its own (one that loaded no state would otherwise keep
                    # the chunk's shared tensor, and step() would advance it once per such parameter)
                    for p, n in zip(sub, counts):
                        self.state[p]["step"] = torch.tensor(float(n))
