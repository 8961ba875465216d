"""hipGraph capture of the whole train step.

The reference's train_step (kfold_train_adversarial.py:101-136) issues ~600 kernels per step; about 250 of them
are the token-side transformer / heads / optimizer kernels that each run for 3-15 us, and in eager mode the
Python + autograd dispatch of the transformer BACKWARD (≈4 ms of host time) sits on the critical path between
the forward and the long convolution backward.  Capturing zero_grad -> forward -> loss -> backward ->
optimizer.step once and replaying it removes the host from the loop: the kernels, their order, their stream
fork/join (MRI and PET encoders on two streams) and their numerics are exactly those of the eager step.

Measured on MI355X / ROCm 7.2 (B=8, 96^3): the replayed step took 19.7 ms against 18.2 ms eager when this was written (28.0 vs
16.8 ms with the current kernels) — hipGraph node
dispatch costs more than it saves here because the eager step is GPU-bound for 85 % of its length — so bench.py
keeps the eager step as the default and offers --graph.

    step = GraphedTrainStep(net, optimizer, loss_fn, example_inputs=(mri, pet, label))
    loss = step(mri, pet, label)          # copies into the static buffers, replays; returns the loss tensor

With a GradAllReduce-wrapped model (N > 1 ranks) the graph holds forward + backward only; the bucketed RCCL
all-reduce and the optimizer step run eagerly after the replay (collectives are kept out of the capture).
"""
from __future__ import annotations

from typing import Callable, Sequence

import torch


class GraphedTrainStep:
    def __init__(self, net, optimizer, loss_fn: Callable, example_inputs: Sequence[torch.Tensor],
                 warmup: int = 3):
        from .parallel import GradAllReduce
        self.net = net
        self.opt = optimizer
        self.loss_fn = loss_fn
        self.ddp = net if isinstance(net, GradAllReduce) and net.world > 1 else None
        self.static_in = [t.clone() for t in example_inputs]
        dev = self.static_in[0].device
        if dev.type != "cuda":
            raise RuntimeError("GraphedTrainStep needs HIP tensors")
        for g in optimizer.param_groups:
            if "capturable" in g and not g["capturable"] and self.ddp is None:
                raise RuntimeError("construct the optimizer with capturable=True to capture its step")
        net.train()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._eager_body()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        if self.ddp is not None:
            self.ddp.require_sync = False
        with torch.cuda.graph(self.graph):
            self.static_loss = self._captured_body()
        if self.ddp is not None:
            self.ddp.require_sync = True

    def _fwd_bwd(self):
        out = self.net(*self.static_in[:-1])
        loss = self.loss_fn(out, self.static_in[-1])
        loss.backward()
        return loss

    def _eager_body(self):
        self.opt.zero_grad(set_to_none=True)
        if self.ddp is not None:
            self.ddp.require_sync = False
        loss = self._fwd_bwd()
        if self.ddp is not None:
            self.ddp.require_sync = True
            self.ddp.reduce_gradients()
        self.opt.step()
        return loss

    def _captured_body(self):
        loss = self._fwd_bwd()
        if self.ddp is None:
            self.opt.step()
        return loss

    def __call__(self, *inputs):
        for s, t in zip(self.static_in, inputs):
            if s.data_ptr() != t.data_ptr():
                s.copy_(t, non_blocking=True)
        self.graph.replay()
        if self.ddp is not None:
            self.ddp.reduce_gradients()
            self.opt.step()
        return self.static_loss
