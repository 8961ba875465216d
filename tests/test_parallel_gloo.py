"""The N>1 data-parallel path on CPU: world_size-2 gloo process group, the end-of-backward bucket path of
transmf_ad_amd.parallel.GradAllReduce (a wrapped module that is not one of the HIP models has no flat gradient buffers).  The wrapped module is the
CPU oracle of model_ad (tiny configuration), one different minibatch shard per rank; the averaged
gradients must equal the mean of the per-shard single-process gradients (BatchNorm stays per
replica)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn

from oracle import params as P
from oracle import tmf_oracle as O

KW = dict(dim=32, depth=2, heads=4, dim_head=8, mlp_dim=128)
SIZE = (32, 32, 32)


class OracleModel(nn.Module):
    """nn.Module shell around the functional oracle so that hooks / optimizers see parameters."""

    def __init__(self):
        super().__init__()
        self.spec = O.state_spec("model_ad", **KW)
        S = O.to_state(P.init_arrays(self.spec, seed=7), self.spec, requires_grad=False)
        self.names = [k for k, (kind, _s) in self.spec.items() if kind == "param"]
        self.ps = nn.ParameterList([nn.Parameter(S[k]) for k in self.names])
        self.bufs = {k: S[k] for k, (kind, _s) in self.spec.items() if kind == "buffer"}

    def forward(self, mri, pet):
        S = dict(self.bufs)
        S.update({k: p for k, p in zip(self.names, self.ps)})
        k1, k2 = (torch.from_numpy(m) for m in P.make_masks(mri.shape[0]))
        return O.model_ad_forward(S, mri, pet, dim=KW["dim"], depth=KW["depth"], heads=KW["heads"], train=True,
                                  dropout_masks=(k1, k2))


def _shard(rank):
    mri, pet, y = P.make_inputs(2, SIZE, seed=100 + rank)
    return torch.from_numpy(mri), torch.from_numpy(pet), torch.from_numpy(y)


def _local_grads(rank):
    torch.manual_seed(0)
    m = OracleModel()
    mri, pet, y = _shard(rank)
    O.adversarial_loss(*m(mri, pet), y).backward()
    return [p.grad.clone() for p in m.ps]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    from transmf_ad_amd.parallel import GradAllReduce, init_from_env
    r, _l, w = init_from_env("gloo")
    assert (r, w) == (rank, world)
    m = OracleModel()
    if rank == 1:                      # rank-0 broadcast must overwrite this
        with torch.no_grad():
            m.ps[0].add_(1.0)
    net = GradAllReduce(m, bucket_mb=0.05)
    assert len(net.bucket_sizes_bytes) > 3
    mri, pet, y = _shard(rank)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    opt.zero_grad()
    O.adversarial_loss(*net(mri, pet), y).backward()
    grads = [p.grad.clone() for p in m.ps]
    opt.step()
    # second step exercises bucket reset + zero_grad(set_to_none=True)
    opt.zero_grad()
    O.adversarial_loss(*net(mri, pet), y).backward()
    g2 = [p.grad.clone() for p in m.ps]
    # .grad now aliases the reduced buckets; zero_grad(set_to_none=False) keeps those views and the next backward
    # accumulates into them in place — the result must still be the plain averaged gradient
    opt.zero_grad(set_to_none=False)
    O.adversarial_loss(*net(mri, pet), y).backward()
    g3 = [p.grad.clone() for p in m.ps]
    # the collectives of a step: same kinds and byte counts, in the same order, on every rank and in every step
    order3 = (list(net.last_reduced_kinds), list(net.last_reduced_bytes))
    # two forwards before one backward are refused (both ranks raise: nothing is left half-reduced); a forward whose graph is
    # simply dropped is fine, and the step after a refusal is reduced as usual
    refused = False
    opt.zero_grad()
    try:
        (O.adversarial_loss(*net(mri, pet), y) + O.adversarial_loss(*net(mri, pet), y)).backward()
    except RuntimeError as e:
        refused = "one forward per backward" in str(e)
    opt.zero_grad()
    net(mri, pet)                                                   # dropped graph
    O.adversarial_loss(*net(mri, pet), y).backward()
    g4 = [p.grad.clone() for p in m.ps]
    torch.save(dict(grads=grads, p0=m.ps[0].detach().clone(), g2=g2, g3=g3, g4=g4, refused=refused, order3=order3,
                    unreduced=net.unreduced_gradients()), os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gradient_average(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(tmp_path / f"rank{r}.pt") for r in range(world)]
    torch.set_num_threads(2)                                   # same reduction order as the workers
    want = [sum(gs) / world for gs in zip(*[_local_grads(r) for r in range(world)])]
    for a, b, w in zip(res[0]["grads"], res[1]["grads"], want):
        assert torch.equal(a, b)                              # every rank holds the same averaged gradient
        # fp32 conv weight-gradients carry ~1e-5..1e-3 (of max) summation-order noise on the host
        assert (a - w).abs().max() <= 1e-3 * w.abs().max().clamp_min(1e-6)
    assert torch.equal(res[0]["p0"], res[1]["p0"])             # replicas stay in lock-step after Adam
    for a, b in zip(res[0]["g2"], res[1]["g2"]):
        assert torch.equal(a, b)
    for a, b, c in zip(res[0]["g3"], res[1]["g3"], res[0]["g2"]):
        assert torch.equal(a, b)
        assert (a - c).abs().max() <= 1e-3 * c.abs().max().clamp_min(1e-6)     # same parameters, same data as step 2
    assert res[0]["refused"] and res[1]["refused"] and not res[0]["unreduced"] and not res[1]["unreduced"]
    assert res[0]["order3"] == res[1]["order3"] and set(res[0]["order3"][0]) == {"bucket"}
    for a, b, c in zip(res[0]["g4"], res[1]["g4"], res[0]["g2"]):
        assert torch.equal(a, b)
        assert (a - c).abs().max() <= 1e-3 * c.abs().max().clamp_min(1e-6)


def _plan_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    import transmf_ad_amd as T
    from transmf_ad_amd.parallel import GradAllReduce
    dist.init_process_group("gloo", rank=rank, world_size=world)      # a 1-rank group: only the plan is inspected
    torch.manual_seed(0)
    net = T.model_ad(128, 3, 4, 32, 512, 0.)            # parameter holders only: no forward on the CPU
    ddp = GradAllReduce(net, broadcast_from_rank0=False)
    names = {id(p): n for n, p in net.named_parameters()}
    plan = [[names[id(p)] for p in b.params] for b in ddp._buckets]
    if rank == 0:
        torch.save(dict(sizes=ddp.bucket_sizes_bytes, plan=plan, order=[n for n, _ in net.named_parameters()]),
                   os.path.join(out_dir, "plan.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_bucket_plan_of_the_benchmark_model(tmp_path):
    """model_ad(128, 3, 4, 32, 512) has 4 173 060 fp32 parameters = 16.69 MB of gradients, bucketed in REVERSE
    registration order (the order backward produces gradients: heads, fusion transformer, then the encoders), every parameter
    in exactly one bucket.  Heads + fusion fill one bucket under the 6 MB cap; each encoder's deep blocks (sNet tags conv3 /
    conv4: 95 % of its bytes, complete long before the encoder's backward returns, so that bucket starts early) get a
    bucket of their own and the shallow blocks of BOTH encoders share the last one — 4 buckets."""
    mp.spawn(_plan_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    r = torch.load(tmp_path / "plan.pt")
    sizes, plan, order = r["sizes"], r["plan"], r["order"]
    assert sum(sizes) == 4173060 * 4 and abs(sum(sizes) / 1e6 - 16.69) < 0.01
    assert len(sizes) == 4 and all(s <= 6.0 * (1 << 20) + 3.6e6 for s in sizes)        # a bucket closes once it would exceed the cap
    flat = [n for b in plan for n in b]
    assert sorted(flat) == sorted(order) and len(set(flat)) == len(flat)               # every parameter exactly once
    assert plan[0] == list(reversed(order))[:len(plan[0])] and plan[0][0].startswith("D.")   # heads + fusion: reverse registration order
    assert all(n.startswith(("pet_cnn.conv3.", "pet_cnn.conv4.")) for n in plan[1]) and len(plan[1]) == 16
    assert all(n.startswith(("mri_cnn.conv3.", "mri_cnn.conv4.")) for n in plan[3]) and len(plan[3]) == 16
    assert all(".conv1." in n or ".conv2." in n for n in plan[2]) and len(plan[2]) == 24
    assert sizes[1] == sizes[3] and sizes[1] / (sizes[1] + sizes[2] / 2) > 0.93
    # the bucket that closes latest in backward holds the first-block gradients of both encoders
    assert any(n.startswith("mri_cnn.conv1.") for n in plan[2]) and any(n.startswith("pet_cnn.conv1.") for n in plan[2])


def _unused_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    from transmf_ad_amd.parallel import GradAllReduce, init_from_env
    init_from_env("gloo")
    torch.manual_seed(0)
    class Net(nn.Sequential):
        def forward(self, x):                     # `unused` is registered but takes no part in forward
            return self[2](self[1](self[0](x)))
    m = Net(nn.Linear(4, 4), nn.Linear(4, 4), nn.Linear(4, 2))
    extra = nn.Linear(4, 4)
    m.add_module("unused", extra)
    net = GradAllReduce(m, bucket_mb=0.0001)
    x = torch.randn(3, 4) + rank
    net(x).sum().backward()                       # through the wrapper, as with torch DDP
    ok = extra.weight.grad is None and extra.bias.grad is None and all(p.grad is not None for p in m[0].parameters())
    g0 = m[0].weight.grad.clone()
    torch.save(dict(ok=ok, g0=g0), os.path.join(out_dir, f"u{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_unused_parameters_keep_grad_none(tmp_path):
    """A parameter that receives no gradient keeps .grad = None on every rank (as in a single-process run, so Adam skips
    it in both), while the used ones are averaged."""
    world = 2
    mp.spawn(_unused_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(tmp_path / f"u{r}.pt") for r in range(world)]
    assert all(r["ok"] for r in res)
    assert torch.equal(res[0]["g0"], res[1]["g0"])
