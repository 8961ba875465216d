"""Data-parallel wrapper on the real HIP path: two ranks share the one GPU of the test box (gloo backend with
device tensors — RCCL refuses two ranks on one device), each with its own minibatch shard.  Exercises the
autograd hooks, the staging stream and the two-stream encoders together; the averaged gradients must equal
the mean of single-process per-shard gradients.  The RCCL code path itself (backend "nccl": asynchronous work handles
on the staging stream, event-driven deep buckets, record_stream) runs in a 1-rank group — all a 1-GPU box allows."""
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


def test_deep_block_gradients_carry_an_early_event():
    """tmf_snet_train_bwd records an event behind the last kernel of blocks conv3.0 .. conv4.3; ops.grad_ready_event finds
    it for exactly those gradients (and not for conv1 / conv2, nor for a buffer that is not the one of this backward), and
    behind the event the deep gradients already hold their final values."""
    from transmf_ad_amd import ops
    net = _build().train()
    ops.GRAD_READY_EVENTS.clear()
    ops.TRACK_GRAD_EVENTS = False
    _loss(net, 0).backward()
    assert not ops.GRAD_READY_EVENTS            # nobody consumes them: nothing is published (and nothing is held)
    net.zero_grad()
    ops.TRACK_GRAD_EVENTS = True                # what parallel.GradAllReduce switches on
    _loss(net, 0).backward()
    deep = [p for n, p in net.named_parameters() if ".conv3." in n or ".conv4." in n]
    shallow = [p for n, p in net.named_parameters() if ".conv1." in n or ".conv2." in n]
    assert len(deep) == 2 * 16 and len(shallow) == 2 * 12
    evs = [ops.grad_ready_event(p.grad) for p in deep]
    assert all(e is not None for e in evs) and len({id(e) for e in evs}) == 2          # one event per encoder
    assert all(ops.grad_ready_event(p.grad) is None for p in shallow)
    assert all(ops.grad_ready_event(p.grad) is None for n, p in net.named_parameters() if "_cnn." not in n)
    for e in {id(e): e for e in evs}.values():
        e.synchronize()
    early = [p.grad.clone() for p in deep]
    torch.cuda.synchronize()
    for a, p in zip(early, deep):
        assert torch.equal(a, p.grad)
    assert ops.grad_ready_event(deep[0].grad.clone()) is None                           # another buffer: no event
    ops.GRAD_READY_EVENTS.clear()
    ops.TRACK_GRAD_EVENTS = False


def test_bucket_groups_follow_the_deep_shallow_split(tmp_path):
    """With the wrapper, the deep-block buckets of both encoders wait for their event, not for the producing stream; the
    shared shallow bucket waits for both encoder streams."""
    world = 1
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", TMF_DDP_FORCE="1")
    try:
        from transmf_ad_amd.parallel import GradAllReduce
        dist.init_process_group("gloo", rank=0, world_size=world)
        net = GradAllReduce(_build(), bucket_mb=64.0)
        names = {p: n for n, p in net.module.named_parameters()}
        kinds = []
        for b in net._buckets:
            ns = [names[p] for p in b.params]
            kinds.append("deep" if all(".conv3." in n or ".conv4." in n for n in ns) else
                         "shallow" if all(".conv1." in n or ".conv2." in n for n in ns) else "other")
        assert kinds == ["other", "deep", "shallow", "deep"], kinds
        seen = {}
        orig = net._launch

        def spy(b):
            seen[id(b)] = (len(b.events), len(b.streams))
            return orig(b)
        net._launch = spy
        net.train()
        _loss(net, 0).backward()
        torch.cuda.synchronize()
        for b, k in zip(net._buckets, kinds):
            ev, st = seen[id(b)]
            assert (ev == 1 and st == 0) if k == "deep" else (ev == 0 and st >= 1), (k, ev, st)
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
        os.environ.pop("TMF_DDP_FORCE", None)


def _nccl_one_rank_worker(_rank, port, out_dir, full):
    """3 train steps (Adam, zero_grad(set_to_none=False) from the second on) of the model wrapped in GradAllReduce over a
    1-rank RCCL group, and of the same model unwrapped: saves parameters + gradients of both."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      TMF_DDP_FORCE="1")             # keep the bucket machinery live in a 1-rank group
    import transmf_ad_amd as T
    from transmf_ad_amd.parallel import GradAllReduce
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    res = {}
    for wrapped in (True, False):
        if full:
            torch.manual_seed(3)
            net = T.model_ad(dim=128, depth=3, heads=4, dim_head=32, mlp_dim=512, dropout=0.).to("cuda:0")
            net.fc_cls[3] = torch.nn.Identity(); net.fc_cls[7] = torch.nn.Identity()
            mri, pet, y = (torch.from_numpy(a).to("cuda:0") for a in P.make_inputs(2, (48, 48, 48), seed=5, kind="blobs"))
        else:
            net = _build()
            mri, pet, y = (torch.from_numpy(a).to("cuda:0") for a in P.make_inputs(2, SIZE, seed=100))
        model = GradAllReduce(net) if wrapped else net
        model.train()
        opt = T.optim.Adam(model.parameters(), lr=1e-3)
        ce = torch.nn.functional.cross_entropy
        launched = []
        if wrapped:
            model.timing = True
            orig = model._launch

            def spy(b, _o=orig):
                launched.append((len(b.events), len(b.streams), b.tagged))
                return _o(b)
            model._launch = spy
        for it in range(3):
            opt.zero_grad(set_to_none=(it == 0))
            lo, dm, dp = model(mri, pet)
            (ce(lo, y) + (ce(dm, torch.ones_like(y)) + ce(dp, torch.zeros_like(y))) / 2).backward()
            opt.step()
        torch.cuda.synchronize()
        res[wrapped] = {"params": [p.detach().cpu() for p in net.parameters()], "grads": [p.grad.cpu() for p in net.parameters()],
                        "launched": launched, "exposed": model.exposed_allreduce_ms() if wrapped else None,
                        "nbuckets": len(model._buckets) if wrapped else 0}
    torch.save(res, os.path.join(out_dir, "res.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("full", [False, True], ids=["tiny", "dim128"])
def test_rccl_one_rank_group_matches_unwrapped_model(tmp_path, full):
    """backend "nccl" (= RCCL) with ONE rank on the box's GPU: three Adam steps of the wrapped model are BITWISE the steps of
    the unwrapped model (all-reduce over one rank and the division by 1 are exact), incl. zero_grad(set_to_none=False);
    every bucket was launched every step, the deep-block buckets of the first step behind ONE event and no stream (the event
    path of tmf_snet_train_bwd), everything else behind streams; the exposed all-reduce times are finite (no stream leak)."""
    mp.spawn(_nccl_one_rank_worker, args=(_free_port(), str(tmp_path), full), nprocs=1, join=True)
    res = torch.load(tmp_path / "res.pt", weights_only=False)
    w, u = res[True], res[False]
    for a, b in zip(w["params"], u["params"]):
        assert torch.equal(a, b)
    for a, b in zip(w["grads"], u["grads"]):
        assert torch.equal(a, b)
    assert len(w["launched"]) == 3 * w["nbuckets"] and w["nbuckets"] >= 3
    # step 1 (fresh gradients = views of tmf_snet_train_bwd's flat buffer): the deep buckets wait for ONE event and no
    # stream; steps 2, 3 (set_to_none=False: autograd accumulates IN PLACE into the bucket views, on the producing stream):
    # no event applies to that add, the buckets wait for the streams
    deep = [l for l in w["launched"] if l[2]]
    assert len(deep) == 3 * 2 and all(ev == 1 and st == 0 for ev, st, _t in deep[:2]), deep
    assert all(ev == 0 and st >= 1 for ev, st, _t in deep[2:]), deep
    assert all(ev == 0 and st >= 1 for ev, st, t in w["launched"] if not t)
    assert len(w["exposed"]) == 3 and all(np.isfinite(x) and 0 <= x < 1e3 for x in w["exposed"]), w["exposed"]
