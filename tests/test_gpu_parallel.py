"""Data-parallel wrapper on the real HIP path: two ranks share the one GPU of the test box (gloo backend with
device tensors — RCCL refuses two ranks on one device), each with its own minibatch shard.  Exercises the
autograd hooks, the staging stream and the two-stream encoders together; the averaged gradients must equal
the mean of single-process per-shard gradients."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import params as P
from oracle import tmf_oracle as O

pytestmark = pytest.mark.gpu
KW = dict(dim=32, depth=2, heads=4, dim_head=8, mlp_dim=128)
SIZE = (32, 32, 32)


def _build():
    import transmf_ad_amd as T
    spec = O.state_spec("model_ad", **KW)
    net = T.model_ad(dropout=0., **KW)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in P.init_arrays(spec, seed=7).items()}, strict=True)
    net = net.to("cuda:0")
    net.fc_cls[3] = torch.nn.Identity()          # no RNG in the comparison
    net.fc_cls[7] = torch.nn.Identity()
    return net


def _loss(net, rank):
    mri, pet, y = (torch.from_numpy(a).to("cuda:0") for a in P.make_inputs(2, SIZE, seed=100 + rank))
    lo, dm, dp = net(mri, pet)
    ce = torch.nn.functional.cross_entropy
    return ce(lo, y) + (ce(dm, torch.ones_like(y)) + ce(dp, torch.zeros_like(y))) / 2


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from transmf_ad_amd.parallel import GradAllReduce, init_from_env
    init_from_env("gloo")
    net = GradAllReduce(_build(), bucket_mb=0.05)
    net.train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    for _ in range(2):                       # second pass: bucket reset, set_to_none grads, stream reuse
        opt.zero_grad()
        _loss(net, rank).backward()
    torch.cuda.synchronize()
    torch.save([p.grad.cpu() for p in net.parameters()], os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_one_gpu_gradient_average(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = [torch.load(tmp_path / f"rank{r}.pt") for r in range(world)]
    local = []
    for r in range(world):
        net = _build().train()
        _loss(net, r).backward()
        torch.cuda.synchronize()
        local.append([p.grad.cpu() for p in net.parameters()])
    for a, b, l0, l1 in zip(got[0], got[1], local[0], local[1]):
        assert torch.equal(a, b)
        want = (l0 + l1) / 2
        assert (a - want).abs().max() <= 1e-6 * max(1.0, want.abs().max().item())
