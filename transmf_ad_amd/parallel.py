"""Data-parallel training over the GPUs of one node: one process per GPU, gradients
all-reduced with RCCL (torch.distributed backend "nccl" on ROCm) over xGMI.

The reference is single-process / single-GPU (no DDP, no NCCL anywhere), so
there is no call pattern to mirror; this is sized for the path itself:

  * the replica is whole (4.17 M parameters = 16.7 MB of fp32 gradients for model_ad), each rank
    runs the reference batch of 8 pairs, BatchNorm statistics stay per replica (reference
    semantics at batch 8 — no SyncBN);
  * gradients are packed into a few flat buckets in REVERSE registration order — the order
    backward produces them: heads and fusion transformer first, then conv4 ... conv1 — and
    each bucket's all-reduce is launched asynchronously from a post-accumulate-grad hook the
    moment its last gradient lands: the heads + fusion bucket rides under the encoders' backward;
    an encoder's deep blocks (conv3, conv4: 95 % of its bytes) have a bucket of their own that
    waits for the event tmf_snet_train_bwd records behind them and rides under conv2 / conv1;
    the shallow blocks of both encoders share the small last bucket.  xGMI is point-to-point (≈153 GB/s per link): a ring over 8 GPUs moves
    2*(7/8)*16.7 MB ≈ 29 MB per GPU per step (≈0.2 ms on one link) — few, large buckets keep
    that bandwidth-bound rather than latency-bound;
  * a queued autograd callback waits for the buckets, averages, and re-points ``param.grad`` at views of
    the reduced buckets before ``optimizer.step()`` — the reference's train_step is unchanged.  Per step this
    costs one multi-tensor pack, one all-reduce and one scale per bucket (no per-parameter kernels).
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.distributed as dist
from torch import nn


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class _Bucket:
    def __init__(self, params: List[nn.Parameter]):
        self.params = params
        self.numel = sum(p.numel() for p in params)
        self.offsets = []
        o = 0
        for p in params:
            self.offsets.append(o)
            o += p.numel()
        p0 = params[0]
        self.flat = torch.zeros(self.numel, dtype=p0.dtype, device=p0.device)
        self.pending = len(params)
        self.filled = [False] * len(params)
        self.work = None
        self.streams = {}
        self.events = []
        self.tagged = all(getattr(p, "tmf_bucket_group", (None,))[0] == "sNet deep" for p in params)

    def reset(self):
        self.pending = len(self.params)
        self.filled = [False] * len(self.params)
        self.work = None
        self.streams = {}
        self.events = []


class GradAllReduce(nn.Module):
    """Wrap a module so that ``loss.backward()`` leaves rank-averaged gradients in ``.grad``.

    >>> net = GradAllReduce(model_ad(...).to(device))      # after dist.init_process_group
    >>> out = net(mri, pet); loss.backward(); optimizer.step()
    """

    def __init__(self, module: nn.Module, process_group=None, bucket_mb: float = 6.0,
                 broadcast_from_rank0: bool = True):
        super().__init__()
        from . import ops
        ops.TRACK_GRAD_EVENTS = True         # the encoder nodes publish their early events from now on (ops.GRAD_READY_EVENTS)
        if not dist.is_available() or not dist.is_initialized():
            raise RuntimeError("GradAllReduce needs an initialised torch.distributed process group")
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self._buckets: List[_Bucket] = []
        self._where = {}
        self._callback_queued = False
        self._streams = {}
        self.require_sync = True
        # timing = True: record a pair of events per step around "all buckets reduced" on the staging stream — the time
        # backward's END has to wait for the collectives that did not hide under it (exposed_allreduce_ms())
        self.timing = False
        self._timing_events = []
        # TMF_DDP_FORCE=1 keeps the bucket machinery live in a 1-rank group (to measure its overhead on one GPU)
        self._force = os.environ.get("TMF_DDP_FORCE", "0") == "1"
        # TMF_DDP_EVENTS=0: every bucket waits for the whole producing stream (no early start of the deep-block buckets)
        self._early = os.environ.get("TMF_DDP_EVENTS", "1") != "0"
        self._live = self.world > 1 or self._force
        # a module that runs parts of its backward on streams of its own says so (`tmf_backward_streams(device)` -> the
        # streams besides the caller's): the hooks then need not look up the current stream for every gradient
        self._known_streams = getattr(module, "tmf_backward_streams", None)
        params = [p for p in module.parameters() if p.requires_grad]
        if broadcast_from_rank0 and self.world > 1:
            with torch.no_grad():
                for t in list(module.parameters()) + list(module.buffers()):
                    dist.broadcast(t, 0, group=process_group)
        cap = int(bucket_mb * (1 << 20))
        # Buckets fill in REVERSE registration order (the order backward produces gradients), a new one when the cap would
        # be exceeded.  A module may tag parameters with `tmf_bucket_group`: tagged parameters only share a bucket with
        # their own group (which may be spread over the model).  sNet tags its deep blocks (conv3, conv4: 95 % of an
        # encoder's bytes, complete long before the encoder's backward call returns — ops.GRAD_READY_EVENTS — so their
        # bucket starts its all-reduce under conv2 / conv1) per encoder, and the shallow blocks of ALL encoders as one group
        # (one small bucket at the very end instead of one per encoder).
        plan, sizes, named = [], [], {}
        cur = None
        for p in reversed(params):
            nbytes = p.numel() * p.element_size()
            gk = getattr(p, "tmf_bucket_group", None)
            i = cur if gk is None else named.get(gk)
            if i is not None and (sizes[i] + nbytes > cap or p.dtype != plan[i][0].dtype or p.device != plan[i][0].device):
                i = None
            if i is None:
                plan.append([]); sizes.append(0)
                i = len(plan) - 1
            plan[i].append(p)
            sizes[i] += nbytes
            if gk is None:
                cur = i
            else:
                named[gk] = i
                cur = None                       # an untagged run does not continue across tagged parameters
        for b_ in plan:
            self._add_bucket(b_)
        for p in params:
            p.register_post_accumulate_grad_hook(self._on_grad)

    # -- construction ----------------------------------------------------------------------
    def _add_bucket(self, params):
        b = _Bucket(list(params))
        for i, p in enumerate(b.params):
            self._where[p] = (b, i)
        self._buckets.append(b)

    @property
    def bucket_sizes_bytes(self):
        return [b.numel * b.flat.element_size() for b in self._buckets]

    # -- backward-time machinery -----------------------------------------------------------
    # Gradients are produced on more than one HIP stream (the MRI and PET encoders run on two streams and
    # autograd replays each backward on its forward stream).  All bucket traffic therefore goes through ONE
    # staging stream per device: it waits for the producing stream, packs the gradient, and issues the
    # collective; RCCL orders itself against that stream.
    def _staging(self, device):
        key = (device.type, device.index)
        if key not in self._streams:
            self._streams[key] = torch.cuda.Stream(device=device)
        return self._streams[key]

    def _views(self, b):
        # built once per bucket (the flat buffer lives as long as the wrapper): a bucket of ~100 small parameters costs
        # ~200 us of Python to slice, and the heads + fusion bucket is packed while the GPU waits for the encoders' backward
        v = getattr(b, "_view_cache", None)
        if v is None:
            v = b._view_cache = [b.flat[o:o + p.numel()].view_as(p) for o, p in zip(b.offsets, b.params)]
        return v

    def _launch(self, b):
        """Pack the bucket with ONE multi-tensor copy (all of its gradients exist by now) and start its all-reduce.
        Gradients come from more than one stream (MRI / PET encoders): the staging stream waits on every stream a
        hook of this bucket fired on."""
        views = self._views(b)
        grads, dst = [], []
        for i, p in enumerate(b.params):
            if b.filled[i] and p.grad is not None:
                g = p.grad
                grads.append(g if g.shape == p.shape else g.reshape(p.shape))
                dst.append(views[i])
            else:                               # no gradient this pass: contribute zeros
                views[i].zero_()
        if grads:
            torch._foreach_copy_(dst, grads)
        b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _on_grad(self, p: torch.Tensor):
        # Runs once per parameter in the middle of backward, and the start of backward (heads, fusion: ~100 small
        # parameters) is launch-bound: every microsecond here is a microsecond of idle GPU.  So a hook only counts; what a
        # bucket has to wait for is worked out once, when its last gradient lands.
        if not (self._live and self.require_sync):
            return
        b, i = self._where[p]
        if b.filled[i]:
            return
        b.filled[i] = True
        b.pending -= 1
        if not self._callback_queued:
            self._callback_queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self._finalize)
        if self._known_streams is None and p.grad.is_cuda:          # a module of unknown stream use: track per gradient
            cur = torch.cuda.current_stream(p.device)
            b.streams[cur.cuda_stream] = cur
        if b.pending == 0:
            self._complete(b, p)

    def _complete(self, b, p):
        if not p.grad.is_cuda:
            self._launch(b)
            return
        dev = p.device
        st = self._staging(dev)
        ev = None
        if self._early and b.tagged:
            # a tagged bucket's gradients come out of ONE backward call that recorded ONE event behind them (sNet deep
            # blocks): take the event when the first and the last gradient of the bucket both carry it
            from . import ops
            g0, g1 = b.params[0].grad, b.params[-1].grad
            e0 = ops.grad_ready_event(g0) if g0 is not None else None
            e1 = ops.grad_ready_event(g1) if g1 is not None else None
            if e0 is not None and e0 is e1:
                ev = e0
        if ev is not None:
            b.events.append(ev)
            st.wait_event(ev)
        else:
            cur = torch.cuda.current_stream(dev)
            b.streams[cur.cuda_stream] = cur
            if self._known_streams is not None:
                for s_ in self._known_streams(dev):
                    b.streams[s_.cuda_stream] = s_
            for s_ in b.streams.values():
                st.wait_stream(s_)
        with torch.cuda.stream(st):
            self._launch(b)
            for q in b.params:
                if q.grad is not None:
                    q.grad.record_stream(st)

    def _finalize(self):
        self._callback_queued = False
        try:
            from . import ops
            ops.GRAD_READY_EVENTS.clear()
        except Exception:                        # pragma: no cover - CPU-only use without the HIP library
            pass
        dev = self._buckets[0].flat.device
        cuda = dev.type == "cuda"
        st = self._staging(dev) if cuda else None
        ctx = torch.cuda.stream(st) if cuda else _NullCtx()
        if cuda:
            st.wait_stream(torch.cuda.current_stream(dev))
        with ctx:
            for b in self._buckets:
                if b.work is None:        # some parameter got no gradient this pass
                    self._launch(b)
            ev0 = ev1 = None
            if cuda and self.timing:
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record(st)
            for b in self._buckets:
                b.work.wait()
                b.flat.div_(self.world)
            if ev0 is not None:
                ev1.record(st)
                self._timing_events.append((ev0, ev1))
        if cuda:
            torch.cuda.current_stream(dev).wait_stream(st)
        # .grad becomes a VIEW of the reduced bucket (no copy back); the next zero_grad() drops it and the next
        # backward's pack — stream-ordered after the optimizer step that reads these views — refills the bucket.
        # A parameter that received no gradient in this backward keeps ``.grad = None`` — exactly what a single-process
        # run leaves, so Adam skips it there and here alike (its zero-filled slice still rode in the all-reduce).  As
        # with torch DDP(find_unused_parameters=False) the set of used parameters must be the same on every rank.
        for b in self._buckets:
            for i, (p, v) in enumerate(zip(b.params, self._views(b))):
                if b.filled[i]:
                    p.grad = v
            b.reset()

    def exposed_allreduce_ms(self):
        """Per recorded step: milliseconds between the end of backward's compute and the last bucket being reduced and
        scaled (what the collectives cost on top of backward).  Synchronises; clears the record."""
        if not self._timing_events:
            return []
        self._timing_events[-1][1].synchronize()
        out = [a.elapsed_time(b) for a, b in self._timing_events]
        self._timing_events = []
        return out

    def reduce_gradients(self):
        """Synchronous form (no overlap): all-reduce every bucket from the gradients currently in ``.grad``.
        Used after a hipGraph replay of forward+backward, where the autograd hooks do not run."""
        if self.world == 1:
            return
        for b in self._buckets:
            b.filled = [p.grad is not None for p in b.params]
            self._launch(b)
        for b in self._buckets:
            b.work.wait()
            b.flat.div_(self.world)
            # copy back INTO the existing .grad tensors (a captured graph owns them and rewrites them on replay)
            views = self._views(b)
            dst = [p.grad for p in b.params if p.grad is not None]
            src = [v for p, v in zip(b.params, views) if p.grad is not None]
            if dst:
                torch._foreach_copy_(dst, src)
            for p, v in zip(b.params, views):
                if p.grad is None:
                    p.grad = v.clone()
            b.reset()

    # -- nn.Module surface -------------------------------------------------------------------
    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def state_dict(self, *args, **kwargs):          # checkpoints interchange with the bare module
        return self.module.state_dict(*args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        return self.module.load_state_dict(*args, **kwargs)


def init_from_env(backend: Optional[str] = None):
    """Initialise the default process group from the torchrun environment
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT).  Returns (rank, local_rank, world)."""
    import os
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world
