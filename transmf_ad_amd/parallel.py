"""Data-parallel training over the GPUs of one node: one process per GPU, gradients
all-reduced with RCCL (torch.distributed backend "nccl" on ROCm) over xGMI.

The reference is single-process / single-GPU (no DDP, no NCCL anywhere), so
there is no call pattern to mirror; this is sized for the path itself:

  * the replica is whole (4.17 M parameters = 16.7 MB of fp32 gradients for model_ad), each rank
    runs the reference batch of 8 pairs, BatchNorm statistics stay per replica (reference
    semantics at batch 8 — no SyncBN);
  * every whole-pass autograd node of the models (heads, fusion transformer, each encoder) already writes ALL of its
    parameter gradients into ONE flat buffer and hands autograd views of it (ops.SNetTrain / FusionTrain / HeadsAD /
    HeadsCNN).  The wrapper is told about each buffer at the end of the node's backward (ops.add_flat_grad_consumer) and
    all-reduces it IN PLACE, asynchronously, on a staging stream: no per-parameter hooks, no pack copies, and
    ``param.grad`` simply is a view of the reduced buffer.  Order = the order backward produces them: heads (1.3 MB) and
    fusion (4.7 MB) first — they ride under the encoders' backward; an encoder's deep blocks (conv3, conv4: 95 % of its
    bytes) are final at the event tmf_snet_train_bwd records behind them and ride under conv2 / conv1; only the shallow
    blocks of the two encoders (0.33 MB each) are reduced after backward's last kernel.  xGMI is
    point-to-point (≈153 GB/s per link): a ring over 8 GPUs moves 2*(7/8)*16.7 MB ≈ 29 MB per GPU per step (≈0.2 ms on
    one link) — six collectives per step, four of them large;
  * anything that does not arrive that way — a module on the op-per-launch path, a wrapped module that is not one of ours
    (the CPU tests wrap the oracle), a backward that accumulates into existing ``.grad`` tensors
    (``zero_grad(set_to_none=False)``) — is reduced at the END of backward from ``param.grad`` through a few flat buckets
    (reverse registration order, ``bucket_mb`` each); correct for every module, without overlap;
  * one tensor hook per model output queues the end-of-backward callback (three hooks for model_ad instead of one per
    parameter); the callback waits for the collectives, averages, and hands the step back to the caller's stream — the
    reference's train_step is unchanged.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.distributed as dist
from torch import nn


_PROF = os.environ.get("TMF_DDP_PROFILE", "0") == "1"       # host-time accounting of the wrapper's callbacks (diagnosis)
_PROF_T = {}


def _prof_report():
    for k, v in sorted(_PROF_T.items(), key=lambda kv: str(kv[0])):
        v2 = v[len(v) // 2:]
        print(f"[ddp profile] {k}: {len(v)} calls, mean of the later half {sum(v2) / max(len(v2), 1) * 1e6:.1f} us", flush=True)


if _PROF:
    import atexit
    atexit.register(_prof_report)


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class _Bucket:
    """Fallback path: a flat buffer for the gradients of `params`, allocated on first use."""

    def __init__(self, params: List[nn.Parameter]):
        self.params = params
        self.numel = sum(p.numel() for p in params)
        self.offsets = []
        o = 0
        for p in params:
            self.offsets.append(o)
            o += p.numel()
        self._flat = None
        self._views = None

    @property
    def nbytes(self):
        return self.numel * self.params[0].element_size()

    def flat(self):
        if self._flat is None:
            p0 = self.params[0]
            self._flat = torch.zeros(self.numel, dtype=p0.dtype, device=p0.device)
        return self._flat

    def views(self):
        # built once per bucket (the flat buffer lives as long as the wrapper): ~100 small parameters cost ~200 us to slice
        if self._views is None:
            f = self.flat()
            self._views = [f[o:o + p.numel()].view_as(p) for o, p in zip(self.offsets, self.params)]
        return self._views


class GradAllReduce(nn.Module):
    """Wrap a module so that ``loss.backward()`` leaves rank-averaged gradients in ``.grad``.

    >>> net = GradAllReduce(model_ad(...).to(device))      # after dist.init_process_group
    >>> out = net(mri, pet); loss.backward(); optimizer.step()

    The forward MUST go through the wrapper (``net(...)``, not ``net.module(...)``): there are no per-parameter hooks, the
    end-of-backward reduction is queued by a hook on the wrapper's outputs (or by a whole-pass node reporting its gradient
    buffer).  A backward over a graph built from the inner module leaves the gradients un-reduced; ``unreduced_gradients()``
    tells, ``reduce_gradients()`` reduces them explicitly.  ONE forward per backward (the reference's train_step never does
    anything else; torch's DDP supports more, this wrapper does not).  What is refused and what is not: a SINGLE backward that
    spans the graphs of two forwards (``(f(net(a)) + f(net(b))).backward()``) raises — inside that backward, when the hook of
    the older forward's outputs fires, i.e. after the younger forward's graph has run and its ranges have been handed to the
    reduction: the gradients left behind by that exception are PARTLY REDUCED and must be zeroed (``zero_grad()``) before the
    next step.  Two forwards whose backwards run one after the other (f1, f2, backward(f1), backward(f2)) are not detected: each
    backward reduces its own graph's gradients, and the second one ACCUMULATES into ``.grad`` tensors that are already
    rank-averaged — the result is the sum of two averaged gradients (as with torch's DDP), through the bucket path.
    """

    def __init__(self, module: nn.Module, process_group=None, bucket_mb: float = 6.0,
                 broadcast_from_rank0: bool = True):
        super().__init__()
        if not dist.is_available() or not dist.is_initialized():
            raise RuntimeError("GradAllReduce needs an initialised torch.distributed process group")
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self._buckets: List[_Bucket] = []
        self._callback_queued = False
        self._streams = {}
        self.require_sync = True
        # timing = True: record a pair of events per step around "all collectives done" on the staging stream — the time
        # backward's END has to wait for the collectives that did not hide under it (exposed_allreduce_ms())
        self.timing = False
        self._timing_events = []
        # TMF_DDP_FORCE=1 keeps the machinery live in a 1-rank group (to measure its overhead on one GPU)
        self._force = os.environ.get("TMF_DDP_FORCE", "0") == "1"
        # TMF_DDP_EVENTS=0: an encoder's deep-block range waits for the whole producing stream like everything else
        self._early = os.environ.get("TMF_DDP_EVENTS", "1") != "0"
        # TMF_DDP_INPLACE=0: ignore the nodes' flat gradient buffers — everything goes through the end-of-backward buckets
        self._inplace_ok = os.environ.get("TMF_DDP_INPLACE", "1") != "0"
        self._live = self.world > 1 or self._force
        # a module that runs parts of its backward on streams of its own says so (`tmf_backward_streams(device)` -> the
        # streams besides the caller's); without it the end-of-backward reduction waits for the caller's stream only
        self._known_streams = getattr(module, "tmf_backward_streams", None)
        params = [p for p in module.parameters() if p.requires_grad]
        self._params = params
        self._by_ptr = None                     # data pointer -> Parameter, rebuilt when a lookup misses (.to(), load)
        if broadcast_from_rank0 and self.world > 1:
            with torch.no_grad():
                for t in list(module.parameters()) + list(module.buffers()):
                    dist.broadcast(t, 0, group=process_group)
        cap = int(bucket_mb * (1 << 20))
        # Fallback buckets fill in REVERSE registration order (the order backward produces gradients), a new one when the cap
        # would be exceeded.  A module may tag parameters with `tmf_bucket_group`: tagged parameters only share a bucket with
        # their own group (sNet tags its deep blocks per encoder and the shallow blocks of ALL encoders as one group).
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
            self._buckets.append(_Bucket(list(b_)))
        # this backward's in-place reductions: (flat, (parameters, view offsets, base address), [work, ...], deferred ranges)
        self._inplace = []
        self._pending = []
        self._events = []
        # "now" reports (an encoder's deep range, final at an event in the middle of its backward) seen in this backward / in
        # the previous one: the host enqueues nothing until the LAST expected one has come in (see _tmf_flat_grads)
        self._now_seen = 0
        self._now_expected = 1
        self._nodes = {}                        # parameter addresses of a node -> (its Parameters, their view offsets)
        self._covered = set()
        self._checked_steps = 0
        self._fwd_seq = 0                       # forwards through the wrapper so far; _bw_seq: the one the running backward belongs to
        self._bw_seq = None
        # RCCL / NCCL average inside the collective (ncclAvg): no division launch afterwards; gloo sums, the wrapper divides
        self._avg = dist.get_backend(process_group) == "nccl" and os.environ.get("TMF_DDP_AVG", "1") != "0"
        self._op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        self.last_reduced_bytes = []            # bytes of every collective of the last backward, in launch order
        self.last_reduced_kinds = []            # "event" | "stream" | "end" (in place) | "bucket" per collective
        if self._live and self._inplace_ok:
            from . import ops
            ops.add_flat_grad_consumer(self)    # held weakly: nothing is published once the wrapper is gone

    @property
    def bucket_sizes_bytes(self):
        """Sizes of the fallback buckets (the end-of-backward path)."""
        return [b.nbytes for b in self._buckets]

    # -- backward-time machinery -----------------------------------------------------------
    # Gradients are produced on more than one HIP stream (the MRI and PET encoders run on two streams and
    # autograd replays each backward on its forward stream).  All collective traffic therefore goes through ONE
    # staging stream per device: it waits for the producing stream (or event) and issues the collective; RCCL orders
    # itself against that stream.
    def _staging(self, device):
        key = (device.type, device.index)
        if key not in self._streams:
            self._streams[key] = torch.cuda.Stream(device=device)
        return self._streams[key]

    def _param_of(self, ptr):
        m = self._by_ptr
        if m is None or ptr not in m:
            m = self._by_ptr = {p.data_ptr(): p for p in self._params}
        return m.get(ptr)

    def _on_backward_start(self, _grad, seq=None):
        """Tensor hook on the wrapped module's outputs (tagged with the number of the forward that produced them): the first
        gradient of a backward pass queues the end-of-backward callback.  Nothing else happens per gradient."""
        if self._live and self.require_sync:
            if self._bw_seq is None:
                self._bw_seq = seq
            elif seq != self._bw_seq:
                # Outputs of TWO forwards in one backward: the gradients of the two graphs are summed inside autograd's input
                # buffers (in place, into the first node's flat buffer where it can) — the in-place path would hand a buffer to RCCL
                # that is still being added to, and the end-of-backward path was seen to copy stale values on its first use in this
                # pattern.  Not needed by the reference's train_step (one forward, one backward): refused loudly instead of reduced
                # wrongly.  (A forward whose graph is simply dropped is fine: its hooks never fire.)
                raise RuntimeError(
                    "GradAllReduce: this backward spans two forwards through the wrapper — run one forward per backward "
                    "(accumulate over micro-batches with zero_grad(set_to_none=False) between complete steps instead)")
            if not self._callback_queued:
                self._callback_queued = True
                torch.autograd.Variable._execution_engine.queue_callback(self._finalize)
        return None

    def tmf_flat_grads(self, flat, param_ptrs, views, segments):
        if not _PROF:
            return self._tmf_flat_grads(flat, param_ptrs, views, segments)
        import time
        t0 = time.perf_counter()
        r = self._tmf_flat_grads(flat, param_ptrs, views, segments)
        _PROF_T.setdefault(len(param_ptrs), []).append(time.perf_counter() - t0)
        return r

    def _tmf_flat_grads(self, flat, param_ptrs, views, segments):
        """ops._publish_flat_grads: a whole-pass node finished enqueueing its backward; `views[i]` (a view of `flat`) is the
        gradient of the parameter at `param_ptrs[i]`.  If every one of them is ours and autograd will ADOPT the views (no
        ``.grad`` to accumulate into), all-reduce the ranges of `flat` in place, each behind its event / the producing
        stream."""
        if not (self._live and self.require_sync and flat.is_cuda):
            return
        # This runs in the launch-bound start of backward for the heads and the fusion block (every microsecond here is a
        # microsecond of idle GPU: a Python loop over the fusion block's 84 parameters was 0.17 ms per step), so what does
        # not change from step to step — which Parameter each view belongs to, where the view sits in the buffer — is
        # worked out once per node and cached under the node's parameter addresses.
        key = tuple(param_ptrs)
        ent = self._nodes.get(key)
        if ent is None:
            plist, offs = [], []
            for ptr, v in zip(param_ptrs, views):
                if ptr is None or v is None:
                    continue
                p = self._param_of(ptr)
                if p is None:
                    plist = None                # not ours (or a parameter whose storage was replaced: the key changes with it)
                    break
                plist.append(p)
                offs.append(v.data_ptr() - flat.data_ptr())     # (only the ADDRESS of a view is ever kept: a kept reference
                #                                                  would make autograd copy the view instead of adopting it)
            ent = self._nodes[key] = (plist, offs)
        plist, offs = ent
        if not plist:
            return
        # autograd ADOPTS the views only where there is no .grad to accumulate into: zero_grad(set_to_none=False) keeps every
        # .grad, so the first and the last parameter of the node tell (a partly zeroed model takes the error of _finalize)
        if plist[0].grad is not None and plist[0] not in self._covered:
            return                              # accumulation into existing .grad tensors: end-of-backward path
        if plist[-1].grad is not None and plist[-1] not in self._covered:
            return
        if plist[0] in self._covered or plist[-1] in self._covered:
            # A SECOND gradient for these parameters in ONE backward (a module used twice in one graph; two forwards are refused
            # earlier, in _on_backward_start): their first buffer is already queued or in flight on the staging stream and
            # AccumulateGrad is about to add this gradient into it in place on the producing stream — a race with RCCL that would
            # leave avg(g1) + local g2.  The address check of _finalize cannot see it (in-place accumulation keeps the pointer),
            # so it is an error here.
            raise RuntimeError(
                "GradAllReduce: a whole-pass node produced a second gradient for parameters whose gradient buffer was already "
                "handed to the collective in this backward (the module is used twice in one graph) — not supported by the "
                "in-place reduction")
        base = flat.data_ptr()
        params = (plist, offs, base)
        if not self._callback_queued:           # (a backward started from an output the wrapper never saw)
            self._callback_queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self._finalize)
        dev = flat.device
        st = self._staging(dev)
        works, deferred = [], []
        # WHEN to hand a range to RCCL (the node says: ops._publish_flat_grads).  Enqueueing a collective costs the host
        # ~0.1-0.2 ms, and the start of backward (heads, fusion block) is launch-bound: a collective enqueued there is that much
        # idle GPU (measured: 0.25 + 0.22 ms of gaps behind the heads' and the fusion block's backward).  So a "next" range
        # (heads, fusion) only gets an event behind its last kernel and waits in `_pending`; it goes out when the first "now"
        # range arrives (an encoder's deep blocks, final at the event recorded in the middle of its backward) — by then that
        # encoder's whole backward, milliseconds of GPU work, is queued and the host is free —, or at the end of backward.  An
        # "end" range (an encoder's shallow blocks) would make the in-order staging stream wait for this encoder's whole backward
        # with the OTHER encoder's deep range queued behind it: it goes out at the end of backward.
        #
        # ... and with TWO encoders the first one's report comes BETWEEN the two backward calls: three collective enqueues there
        # (heads, fusion, its own deep range: ~0.3 ms of host time) keep the second encoder's kernels — the other stream of the
        # step — from being launched, and the GPU runs one stream for that long (round 5, 1-rank RCCL group: +4 % on an 8.5 ms
        # step, more than the end-of-backward buckets cost).  So a "now" range that is not the LAST one expected (as many as the
        # previous backward had: the same on every rank) only joins `_pending` with its event; the last one flushes everything.
        has_now = self._early and any(when == "now" and ev is not None and stop > start for start, stop, ev, when in segments)
        last_now = False
        if has_now:
            self._now_seen += 1
            last_now = self._now_seen >= self._now_expected
            if last_now:
                self._flush_pending(st)
        for start, stop, ev, when in segments:
            if stop <= start:
                continue
            if when == "now" and ev is not None and self._early:
                if last_now:
                    st.wait_event(ev)
                    with torch.cuda.stream(st):
                        works.append(self._all_reduce(flat[start:stop]))
                    self.last_reduced_bytes.append((stop - start) * flat.element_size())
                    self.last_reduced_kinds.append("event")
                else:
                    self._pending.append((flat, start, stop, ev, works, "event"))
            elif when == "next":
                if ev is None:                                   # final behind what the current stream holds now
                    k = len(self._pending)                      # (events are re-used from step to step: creating one is ~10 us)
                    while k >= len(self._events):
                        self._events.append(torch.cuda.Event())
                    ev = self._events[k]
                    ev.record(torch.cuda.current_stream(dev))
                self._pending.append((flat, start, stop, ev, works, "stream"))
            else:                                               # "end" (and "now" with TMF_DDP_EVENTS=0)
                deferred.append((start, stop))
        # (no record_stream on the buffer: it lives until the next zero_grad — .grad holds views of it — and by then the
        #  caller's stream has joined the staging stream in _finalize, so the allocator's ordinary stream-ordered re-use is
        #  safe; record_stream would park the block behind an event query at every later allocation)
        self._covered.update(plist)
        self._inplace.append((flat, params, works, deferred))

    def _all_reduce(self, t):
        return dist.all_reduce(t, op=self._op, group=self.group, async_op=True)

    def _flush_pending(self, st):
        """Start the collectives of the buffers that reported in without an event range (heads, fusion), each behind the
        event recorded at the end of its backward."""
        for flat, start, stop, done, works, kind in self._pending:
            st.wait_event(done)
            with torch.cuda.stream(st):
                works.append(self._all_reduce(flat[start:stop]))
            self.last_reduced_bytes.append((stop - start) * flat.element_size())
            self.last_reduced_kinds.append(kind)
        self._pending = []

    def _reduce_rest(self, st):
        """End-of-backward path: every parameter with a gradient that was not reduced in place, through the flat buckets."""
        launched = []
        for b in self._buckets:
            todo = [(i, p) for i, p in enumerate(b.params) if p.grad is not None and p not in self._covered]
            if not todo:
                continue
            views = b.views()
            src, dst = [], []
            have = {i for i, _p in todo}
            for i, p in todo:
                g = p.grad
                if g.data_ptr() == views[i].data_ptr() and g.shape == p.shape:
                    continue                    # .grad already IS the bucket view (in-place accumulation since the last step)
                src.append(g if g.shape == p.shape else g.reshape(p.shape))
                dst.append(views[i])
            for i in range(len(b.params)):
                if i not in have:
                    views[i].zero_()            # no gradient this pass: contribute zeros
            if src:
                torch._foreach_copy_(dst, src)
            work = self._all_reduce(b.flat())
            self.last_reduced_bytes.append(b.nbytes)
            self.last_reduced_kinds.append("bucket")
            launched.append((b, todo, work))
        return launched

    def _finalize(self):
        if not _PROF:
            return self._finalize_impl()
        import time
        t0 = time.perf_counter()
        self._finalize_impl()
        _PROF_T.setdefault("finalize", []).append(time.perf_counter() - t0)

    def _reset_backward_state(self):
        """Forget everything a backward pass left behind (also after one that threw): no stale work handles, no stale set of
        in-place parameters, and the next backward queues its own end-of-backward callback."""
        self._inplace = []
        self._pending = []
        self._covered = set()
        self._callback_queued = False
        self._bw_seq = None
        if self._now_seen:                       # (what the next backward may expect; a backward without such a report changes nothing)
            self._now_expected = self._now_seen
        self._now_seen = 0

    def _finalize_impl(self):
        try:
            self._finalize_body()
            self._finalized_since_forward = True
        finally:
            self._reset_backward_state()

    def _finalize_body(self):
        self._callback_queued = False
        first = self._params[0] if self._params else None
        dev = first.device if first is not None else torch.device("cpu")
        cuda = dev.type == "cuda"
        st = self._staging(dev) if cuda else None
        ctx = torch.cuda.stream(st) if cuda else _NullCtx()
        if cuda:
            st.wait_stream(torch.cuda.current_stream(dev))
            if self._known_streams is not None:
                for s_ in self._known_streams(dev):
                    st.wait_stream(s_)
        if cuda:
            self._flush_pending(st)              # (no encoder reported in — a module without one —, or fewer than the last backward had)
        with ctx:
            rest = self._reduce_rest(st)
            ev0 = ev1 = None
            if cuda and self.timing:
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record(st)
            flats = []
            for flat, _params, works, deferred in self._inplace:          # (the staging stream waited for every stream above)
                for start, stop in deferred:
                    works.append(self._all_reduce(flat[start:stop]))
                    self.last_reduced_bytes.append((stop - start) * flat.element_size())
                    self.last_reduced_kinds.append("end")
            for flat, _params, works, _d in self._inplace:
                for w in works:
                    w.wait()
                flats.append(flat)
            for b, _todo, work in rest:
                work.wait()
                flats.append(b.flat())
            if flats and not self._avg:           # (RCCL averages inside the collective: no scaling launch)
                torch._foreach_div_(flats, float(self.world))
            if ev0 is not None:
                ev1.record(st)
                self._timing_events.append((ev0, ev1))
        if cuda:
            torch.cuda.current_stream(dev).wait_stream(st)
        # In place: autograd ADOPTED the node's views as .grad (no kernel: the incoming gradient is kept as it is when there
        # is nothing to accumulate into), so .grad already holds the reduced values.  Had autograd cloned or summed a view
        # instead (a parameter that ALSO receives a gradient from elsewhere in the graph), that kernel would have read the
        # buffer while it was being reduced — not recoverable afterwards, so it is an error, not a silent wrong gradient.
        # Every parameter is checked in the wrapper's first steps (what autograd does with a given graph does not change
        # from step to step), afterwards the first and last of each buffer.
        full = self._checked_steps < 3
        self._checked_steps += 1
        for _flat, (plist, offs, base), _works, _d in self._inplace:
            for i in (range(len(plist)) if full else (0, len(plist) - 1)):
                p = plist[i]
                if p.grad is None or p.grad.data_ptr() != base + offs[i]:
                    raise RuntimeError(
                        "GradAllReduce: a gradient view of a whole-pass node was not adopted as .grad (the parameter also "
                        "receives a gradient from elsewhere in the graph?) — set TMF_DDP_INPLACE=0 to reduce everything at "
                        "the end of backward instead")
        # Buckets: .grad becomes a VIEW of the reduced bucket (no copy back); the next zero_grad() drops it, or a
        # zero_grad(set_to_none=False) keeps it and the next backward accumulates into the bucket in place.  A parameter
        # that received no gradient keeps ``.grad = None`` — exactly what a single-process run leaves, so Adam skips it
        # there and here alike.  As with torch DDP(find_unused_parameters=False) the set of used parameters must be the
        # same on every rank.
        for b, todo, _work in rest:
            views = b.views()
            for i, p in todo:
                p.grad = views[i]

    def _begin_backward_bookkeeping(self):
        self._reset_backward_state()            # (a previous backward that threw mid-way must not leak into this one)
        self._finalized_since_forward = False
        self.last_reduced_bytes = []
        self.last_reduced_kinds = []

    def unreduced_gradients(self):
        """True when gradients exist that no collective has seen since the last forward through the wrapper — a backward whose
        graph was built by calling the INNER module (no output hook, no whole-pass node) never queues the end-of-backward
        reduction.  ``reduce_gradients()`` is the explicit form for such callers; optimizers may assert on this."""
        return (self._live and self.require_sync and not getattr(self, "_finalized_since_forward", True)
                and any(p.grad is not None for p in self._params))

    def exposed_allreduce_ms(self):
        """Per recorded step: milliseconds between the end of backward's compute and the last collective being done and
        scaled (what the collectives cost on top of backward).  Synchronises; clears the record."""
        if not self._timing_events:
            return []
        self._timing_events[-1][1].synchronize()
        out = [a.elapsed_time(b) for a, b in self._timing_events]
        self._timing_events = []
        return out

    def reduce_gradients(self):
        """Synchronous form (no overlap): all-reduce every bucket from the gradients currently in ``.grad``.
        For callers that run forward + backward outside autograd's hooks (e.g. a replayed graph)."""
        if self.world == 1:
            return
        for b in self._buckets:
            views = b.views()
            src = [p.grad for p in b.params if p.grad is not None]
            dst = [v for p, v in zip(b.params, views) if p.grad is not None]
            for p, v in zip(b.params, views):
                if p.grad is None:
                    v.zero_()
            if src:
                torch._foreach_copy_(dst, src)
            dist.all_reduce(b.flat(), op=self._op, group=self.group)
            if not self._avg:
                b.flat().div_(self.world)
            # copy back INTO the existing .grad tensors (a captured graph owns them and rewrites them on replay)
            if src:
                torch._foreach_copy_(src, dst)
            for p, v in zip(b.params, views):
                if p.grad is None:
                    p.grad = v.clone()

    # -- nn.Module surface -------------------------------------------------------------------
    def forward(self, *args, **kwargs):
        out = self.module(*args, **kwargs)
        if self._live and self.require_sync and torch.is_grad_enabled():
            self._begin_backward_bookkeeping()
            self._fwd_seq += 1
            seq = self._fwd_seq
            for t in (out if isinstance(out, (tuple, list)) else (out,)):
                if isinstance(t, torch.Tensor) and t.requires_grad:
                    t.register_hook(lambda g, seq=seq: self._on_backward_start(g, seq))
        return out

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
