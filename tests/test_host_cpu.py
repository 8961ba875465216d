"""CPU-side checks (no GPU needed): the C ABI library loads and exports every symbol the header
declares, the Python boundary mirrors the reference's module interface, and the product path
refuses to run without a HIP device (no silent fallback)."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

from _golden import Golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "tmf_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(tmf_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from transmf_ad_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "libtmf_hip.so not built (python -m transmf_ad_amd.build)"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/tmf_hip.h but not exported"
    assert sorted(_lib.PROTOTYPES) == names, "ctypes prototypes out of sync with the header"
    assert _lib.load().tmf_version() == 1


def test_queries_and_argument_validation_without_gpu():
    """Size queries and argument checks are pure host code: they must work (and refuse) with no GPU."""
    from transmf_ad_amd import _lib
    assert _lib.query("tmf_conv3d_stat_blocks", 8, 48, 48, 48, 32, 64, 3) == 8 * 12 * 6 * 6
    assert _lib.query("tmf_conv3d_wgrad_workspace_bytes", 8, 48, 48, 48, 32, 64, 3) % (27 * 32 * 64 * 4) == 0
    assert _lib.query("tmf_conv3d_stat_blocks", 0, 4, 4, 4, 8, 8, 3) == 0
    with pytest.raises(_lib.TmfError, match="NULL"):
        _lib.call("tmf_bn_act_pool_fwd", None, None, None, None, 1, 2, 2, 2, 8, 0, 0.01, None)
    with pytest.raises(_lib.TmfError, match="pool"):
        _lib.call("tmf_bn_act_pool_fwd", 16, 16, 16, 16, 1, 2, 2, 2, 8, 7, 0.01, None)
    with pytest.raises(_lib.TmfError, match="aligned"):
        _lib.call("tmf_conv3d_fwd", 4, 16, 16, None, 1, 4, 4, 4, 8, 8, 3, None)


def test_forward_brick_choice_follows_the_occupancy_model():
    """plan_fwd (csrc/conv3d_mfma.hip) picks the brick per launch: the 8-wave 4x8x8 brick for the benchmark sizes (their
    profiles must not move), the 4-wave 4x4x8 brick where 4x8x8 bricks leave a CU with an extra workgroup (the
    reference's 91x109x91 volumes at the 22x27x22 level), 64-channel workgroups for the 11x13x11 level; the statistics
    buffer is sized by the same plan."""
    from transmf_ad_amd import _lib
    name = lambda D, H, W, ci, co: _lib.query("tmf_conv3d_fwd_kernel_name", 8, D, H, W, ci, co, 3).decode()
    big64, big32 = "FwdCfg<3, 16, 1, 2, 8, 1, 4, 8, 8, 3>", "FwdCfg<3, 16, 1, 1, 8, 1, 4, 8, 8, 3>"
    for S in (96, 128):
        assert name(S // 2, S // 2, S // 2, 32, 32) == big32 and name(S // 2, S // 2, S // 2, 64, 32) == big32
        for ci, co in ((32, 64), ):
            assert name(S // 2, S // 2, S // 2, ci, co) == big64
        for ci, co in ((64, 64), (64, 128), (128, 64)):
            assert name(S // 4, S // 4, S // 4, ci, co) == big64
    assert name(12, 12, 12, 128, 256) == name(12, 12, 12, 256, 128) == "FwdCfg<3, 32, 1, 1, 2, 4, 4, 4, 4, 1>"
    assert name(45, 54, 45, 32, 64) == big64
    half = "FwdCfg<3, 32, 1, 2, 4, 1, 4, 4, 8, 1>"
    assert name(22, 27, 22, 64, 64) == name(22, 27, 22, 64, 128) == name(22, 27, 22, 128, 64) == half
    assert _lib.query("tmf_conv3d_stat_blocks", 8, 22, 27, 22, 64, 64, 3) == 8 * 6 * 7 * 3
    assert name(11, 13, 11, 128, 256) == name(11, 13, 11, 256, 128) == "FwdCfg<3, 32, 1, 1, 2, 2, 4, 4, 4, 1>"
    assert _lib.query("tmf_conv3d_stat_blocks", 8, 11, 13, 11, 128, 256, 3) == 8 * 3 * 4 * 3
    # the opt-in register-tiled form: off by default; conv_rt = 1 takes it for volumes <= 24^3 its 6x6x12 bricks tile, 16-channel
    # workgroups where 32-channel ones would be fewer than 512; conv_rt = 2 wherever the bricks fit; the statistics buffer follows
    try:
        _lib.call("tmf_set_option", b"conv_rt", 1)
        assert name(24, 24, 24, 64, 64) == name(24, 24, 24, 64, 128) == "RtCfg<2>" and name(12, 12, 12, 128, 256) == "RtCfg<1>"
        assert name(48, 48, 48, 32, 64) == big64 and name(22, 27, 22, 64, 64) == half and name(24, 24, 24, 64, 40) == big64
        assert _lib.query("tmf_conv3d_stat_blocks", 8, 24, 24, 24, 64, 64, 3) == 8 * 4 * 4 * 2
        _lib.call("tmf_set_option", b"conv_rt", 2)
        assert name(48, 48, 48, 32, 64) == "RtCfg<2>"
    finally:
        _lib.call("tmf_set_option", b"conv_rt", 0)
    assert name(24, 24, 24, 64, 64) == big64


@pytest.mark.parametrize("name", ["ad_tiny", "cnn_tiny", "single_mid", "ad_mid"])
def test_state_dict_keys_and_shapes_match_reference(name):
    """Keys/shapes recorded from the imported reference's state_dict() (fixture meta) == ours, in order."""
    import transmf_ad_amd as T
    g = Golden(name)
    net = {"model_ad": lambda: T.model_ad(dropout=0., **g.kw), "model_CNN_ad": lambda: T.model_CNN_ad(**g.kw),
           "model_single": lambda: T.model_single(g.kw["dim"])}[g.model]()
    sd = net.state_dict()
    ref = [(k, tuple(s)) for k, s in g.meta["keys"]]
    assert [(k, tuple(v.shape)) for k, v in sd.items()] == ref
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in g.arrays().items()}, strict=True)
    n_params = sum(p.numel() for p in net.parameters())
    if name == "ad_mid":
        assert n_params == 4173060        # model_ad(128,3,4,32,512)


def test_reference_init_semantics():
    """mymodel.py:195-202: kaiming-normal(fan_out) conv weights, BN3d gamma=1 / beta=0."""
    import transmf_ad_amd as T
    torch.manual_seed(0)
    net = T.model_ad(128, 3, 4, 32, 512, 0.)
    w = net.mri_cnn.conv2[3].weight
    fan_out = w.shape[0] * 27
    assert abs(w.std().item() / (2.0 / fan_out) ** 0.5 - 1) < 0.05
    bn = net.pet_cnn.conv3[4]
    assert torch.all(bn.weight == 1) and torch.all(bn.bias == 0)
    assert isinstance(net.fc_cls[3], torch.nn.Dropout) and net.fc_cls[3].p == 0.5


def test_no_cpu_fallback():
    import transmf_ad_amd as T
    net = T.model_single(128)
    with pytest.raises(T.TmfError, match="no CPU fallback"):
        net(torch.zeros(1, 1, 16, 16, 16))
    with pytest.raises(T.TmfError):
        T.networks.ops.layer_norm(torch.zeros(2, 8), torch.ones(8), torch.zeros(8))


def test_weight_packing_round_trip():
    from transmf_ad_amd import ops
    w = torch.randn(6, 4, 3, 3, 3)
    p = ops.pack_weight(w)
    assert p.shape == (3, 3, 3, 4, 6) and p[1, 2, 0, 3, 5] == w[5, 3, 1, 2, 0]
    assert torch.equal(ops.unpack_wgrad(p.reshape(27, 4, 6), 6, 4, 3), w)
    d = ops.pack_weight_dgrad(w)          # w'[26-t][co][ci] = w[t][ci][co]
    assert d.shape == (3, 3, 3, 6, 4) and d[2, 0, 1, 5, 3] == w[5, 3, 0, 2, 1]
    # dgrad identity on the host: conv(dz, w') == autograd's input gradient
    x = torch.randn(1, 4, 5, 5, 5, dtype=torch.float64, requires_grad=True)
    wd = w.double()
    dz = torch.randn(1, 6, 5, 5, 5, dtype=torch.float64)
    torch.nn.functional.conv3d(x, wd, padding=1).backward(dz)
    w_as_conv = ops.pack_weight_dgrad(wd).permute(4, 3, 0, 1, 2)     # (Cin, Cout, 3,3,3)
    assert torch.allclose(torch.nn.functional.conv3d(dz, w_as_conv, padding=1), x.grad, atol=1e-12)


def test_revgrad():
    from transmf_ad_amd import revgrad
    x = torch.randn(3, 4, requires_grad=True)
    y = revgrad(x, torch.Tensor([2]))
    assert torch.equal(y, x)
    y.sum().backward()
    assert torch.equal(x.grad, torch.full_like(x, -2.0))


def test_reference_checkpoint_loads_strict():
    """A checkpoint written from the imported reference (tests/golden/make_ref_checkpoint.py, the mapping ignite's
    Checkpoint stores: {'net_model': state_dict}) loads into the drop-in class with strict=True."""
    import transmf_ad_amd as T
    ck = torch.load(os.path.join(ROOT, "tests", "golden", "ref_ckpt_ad_tiny.pt"))
    sd = {k: v.float() if v.dtype == torch.float16 else v for k, v in ck["net_model"].items()}
    net = T.model_ad(dim=32, depth=2, heads=4, dim_head=8, mlp_dim=128, dropout=0.)
    missing, unexpected = net.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    assert int(net.D[1].num_batches_tracked) == 2 and int(net.fc_cls[1].num_batches_tracked) == 1
    # and back: our state_dict is loadable by anything expecting the reference's keys
    assert list(net.state_dict().keys()) == list(sd.keys())


def test_fast_mode_switch_matches_module_train():
    """model.train()/.eval() through the cached flat loop sets exactly what nn.Module.train() sets, also after a
    sub-module is replaced."""
    import transmf_ad_amd as T
    net = T.model_ad(32, 2, 4, 8, 128, 0.)
    net.eval()
    assert not any(m.training for m in net.modules())
    net.train()
    assert all(m.training for m in net.modules())
    net.fc_cls[3] = torch.nn.Dropout(0.25)               # replace a sub-module: the cache must be rebuilt
    net.fc_cls[3].eval()
    net.train()
    assert net.fc_cls[3].training
    net.add_module("extra", torch.nn.Linear(2, 2))
    net.eval()
    assert not net.extra.training and not any(m.training for m in net.modules())


def test_winograd_halo_slot_map_is_a_conflict_free_bijection():
    """csrc/conv3d_wino.hip keeps the 6x10x10 halo of a chunk in LDS as 16-byte slots, slot(voxel, quad) = 2 G + (quad ^ ((hh >> 1) & 1)),
    G = parity class * 96 + (hd >> 1) * 32 + (hh >> 1) * 6 + (hw >> 1).  Restated here: the map is injective into the 1 536 slots
    of a buffer, the two quads of a voxel are neighbours (one 32-byte piece of global memory per DMA lane pair), and for every
    tap the 32 tiles' ds_read_b128 is conflict-free in both 16-lane service groups (64 banks x 4 bytes = 16 slots)."""
    def slot(hd, hh, hw, quad):
        g = ((hd & 1) * 4 + (hh & 1) * 2 + (hw & 1)) * 96 + (hd >> 1) * 32 + (hh >> 1) * 6 + (hw >> 1)
        return 2 * g + (quad ^ ((hh >> 1) & 1))
    seen = set()
    for hd in range(6):
        for hh in range(10):
            for hw in range(10):
                a, b = slot(hd, hh, hw, 0), slot(hd, hh, hw, 1)
                assert a // 2 == b // 2 and a != b and 0 <= a < 1536 and 0 <= b < 1536
                assert a not in seen and b not in seen
                seen.update((a, b))
    assert len(seen) == 1200
    groups = ([0, 1, 2, 3, 12, 13, 14, 15] + list(range(20, 28)), list(range(4, 12)) + [16, 17, 18, 19, 28, 29, 30, 31])
    for dd in range(4):
        for i in range(4):
            for k in range(4):
                for quad in range(2):
                    for lanes in groups:
                        res = set()
                        for tile in lanes:                       # lane l31 = tile: td = l31 >> 4, th = (l31 >> 2) & 3, tw = l31 & 3
                            td, th, tw = tile >> 4, (tile >> 2) & 3, tile & 3
                            res.add(slot(2 * td + dd, 2 * th + i, 2 * tw + k, quad) % 16)
                        assert len(res) == 16, (dd, i, k, quad)


def test_winograd_folded_brick_slot_map_is_a_conflict_free_bijection():
    """The persistent kernel's second geometry (csrc/conv3d_wino.hip PGeom<1>): four samples x 2x2x2 tiles, the 4 x 6x6x6 halo of
    a chunk as slot(sample, voxel, quad) = 2 G + (quad ^ ((hh >> 1) & 1)), G = parity class * 144 + sample * 36 + (hd >> 1) * 10 +
    (hh >> 1) * 3 + (hw >> 1): injective into the 2 304 slots of a buffer, and for every tap the 32 tiles' ds_read_b128 (lane
    l31 = sample * 8 + td * 4 + th * 2 + tw) is conflict-free in both 16-lane service groups."""
    def slot(s, hd, hh, hw, quad):
        g = ((hd & 1) * 4 + (hh & 1) * 2 + (hw & 1)) * 144 + s * 36 + (hd >> 1) * 10 + (hh >> 1) * 3 + (hw >> 1)
        return 2 * g + (quad ^ ((hh >> 1) & 1))
    seen = set()
    for s in range(4):
        for hd in range(6):
            for hh in range(6):
                for hw in range(6):
                    a, b = slot(s, hd, hh, hw, 0), slot(s, hd, hh, hw, 1)
                    assert a // 2 == b // 2 and a != b and 0 <= a < 2304 and 0 <= b < 2304
                    assert a not in seen and b not in seen
                    seen.update((a, b))
    assert len(seen) == 2 * 4 * 216
    groups = ([0, 1, 2, 3, 12, 13, 14, 15] + list(range(20, 28)), list(range(4, 12)) + [16, 17, 18, 19, 28, 29, 30, 31])
    for dd in range(4):
        for i in range(4):
            for k in range(4):
                for quad in range(2):
                    for lanes in groups:
                        res = set()
                        for tile in lanes:
                            s, td, th, tw = tile >> 3, (tile >> 2) & 1, (tile >> 1) & 1, tile & 1
                            res.add(slot(s, 2 * td + dd, 2 * th + i, 2 * tw + k, quad) % 16)
                        assert len(res) == 16, (dd, i, k, quad)


def test_winograd_form_is_the_default_plan_of_the_encoder():
    """tmf_set_option("conv_wino", ..): default 3; the whole-encoder plan (tmf_snet_saved_bytes — host arithmetic, no GPU) carries
    the 64-position transformed weights (64 x cin x cout floats per layout) exactly for the layers and directions a mode puts on
    the Winograd kernels, and the weight-gradient workspace holds the kernel's slabs: dw's own [27][cin][cout] per split (+ the
    first-stage sums of a two-stage reduction) with the one-wave-per-SIMD kernel, [splits + groups + 1][blocks][64][32][32] with the other."""
    from transmf_ad_amd import _lib
    lib = _lib.load()
    assert lib.tmf_conv_wino_mode() == 3
    assert lib.tmf_conv3d_wino_ok(8, 32) and not lib.tmf_conv3d_wino_ok(8, 16) and not lib.tmf_conv3d_wino_ok(4, 32)
    assert lib.tmf_conv3d_wgrad_wino_ok(32, 64) and not lib.tmf_conv3d_wgrad_wino_ok(16, 32)
    assert lib.tmf_conv3d_wino_bricks(8, 48, 48, 48) == 8 * 12 * 6 * 6 and lib.tmf_conv3d_wino_bricks(2, 7, 9, 13) == 2 * 2 * 2 * 2
    # statistic partials of the persistent kernel: one row per workgroup = compute unit, whatever the volume (256 without a device)
    rows = lib.tmf_conv3d_wino_stat_blocks(8, 48, 48, 48)
    assert 64 <= rows <= 1024 and lib.tmf_conv3d_wino_stat_blocks(2, 7, 9, 13) == rows and lib.tmf_conv3d_wino_stat_blocks(0, 4, 4, 4) == 0
    assert lib.tmf_conv3d_wino_weight_bytes(32, 64) == 64 * 32 * 64 * (4 + 6)      # the fp32 tensor + its 3-way bf16 split
    # conv2.0 at B = 8, 48^3: one (ci, co) block -> 256 slabs in 16 groups
    assert lib.tmf_wino_p_mode() == 1
    assert lib.tmf_conv3d_wino_bricks(8, 12, 12, 12) == 2 * 3 * 3 * 3               # four samples x 4x4x4 bricks where that is fewer tiles
    # ... but not for a large ragged volume (four samples behind one 32-bit buffer resource): 8 x 81x85x83 would be 2 x 21 x 22 x 21
    # folded bricks against 8 x 21 x 11 x 11 — it keeps the one-sample bricks, which the kernel can address
    assert lib.tmf_conv3d_wino_bricks(8, 81, 85, 83) == 8 * 21 * 11 * 11 and lib.tmf_conv3d_wino_bricks(8, 33, 36, 35) == 2 * 9 * 9 * 9
    assert lib.tmf_conv3d_wgrad_wino_workspace_bytes(8, 48, 48, 48, 32, 32) == (256 + 16) * 27 * 1024 * 4
    # conv4.0 (12^3): 4 pairs of samples x 27 bricks = 108 stages in 8 slabs of 14 per (ci, co) block (18 half bricks per sample: 144 in 8 of 18)
    assert lib.tmf_conv3d_wgrad_wino_workspace_bytes(8, 12, 12, 12, 128, 256) == (8 + 1) * 27 * 128 * 256 * 4
    assert lib.tmf_set_option(b"wino_p", 0) == 0
    assert lib.tmf_conv3d_wino_bricks(8, 12, 12, 12) == 8 * 3 * 2 * 2 and lib.tmf_conv3d_wino_stat_blocks(8, 12, 12, 12) == 8 * 3 * 2 * 2
    assert lib.tmf_conv3d_wgrad_wino_workspace_bytes(8, 48, 48, 48, 32, 32) == (256 + 16 + 1) * 64 * 1024 * 4
    assert lib.tmf_set_option(b"wino_p", 1) == 0
    assert lib.tmf_conv3d_wgrad_wino_workspace_bytes(8, 48, 48, 48, 16, 32) == 0
    desc = _lib.SnetDesc(B=8, D=96, H=96, W=96, dim=128, precision=0, storage_bf16=0)
    desc.momentum[:] = [0.1] * 7
    desc.eps[:] = [1e-5] * 7
    desc.slope[:] = [0.01] * 7
    layers = [(32, 32), (32, 64), (64, 64), (64, 128), (128, 256)]          # the five Cin > 1 3x3x3 blocks of sNet(128)
    try:
        size = {}
        for mode in (0, 1, 2, 3):
            assert lib.tmf_set_option(b"conv_wino", mode) == 0
            size[mode] = lib.tmf_snet_saved_bytes(ctypes.byref(desc)), lib.tmf_snet_bwd_scratch_bytes(ctypes.byref(desc))
        # (a Winograd layout = the fp32 tensor + its exact 3-way bf16 split behind it, 4 + 6 bytes per transformed weight)
        per_layout = sum((64 * 10 - 27 * 4) * ci * co for ci, co in layers)     # every size here is a multiple of 256
        assert size[1][0] - size[0][0] == per_layout                       # data-gradient layouts
        # (the statistic partials — one buffer, sized for the block with the most: rows per workgroup instead of per tile)
        vols = [48, 48, 24, 24, 12]
        c1 = lib.tmf_c1_blocks(8, 96, 96, 96, 32) * 2 * 32 * 4
        part_d = max([c1] + [lib.tmf_conv3d_stat_blocks(8, v, v, v, ci, co, 3) * 2 * co * 4 for v, (ci, co) in zip(vols, layers)])
        part_w = max([c1] + [lib.tmf_conv3d_wino_stat_blocks(8, v, v, v) * 2 * co * 4 for v, (ci, co) in zip(vols, layers)])
        assert size[2][0] - size[1][0] == per_layout + part_w - part_d     # + forward layouts
        assert size[3][0] == size[2][0] and size[3][1] >= size[2][1]       # (the weight-gradient slabs in backward's scratch: dw-sized
        #                                                                     per split with the one-wave kernel — no larger than the direct kernel's)
        assert lib.tmf_set_option(b"conv_wino", 4) != 0
    finally:
        lib.tmf_set_option(b"conv_wino", 3)


def test_algorithm_choice_travels_in_the_descriptor():
    """tmf_snet_desc.flags with TMF_SNET_ALGO: the plan of a call (here: the size of its `saved` workspace, host arithmetic)
    follows the descriptor, not the process options — and without the bit it follows the options; tmf_snet_algo_flags() is the
    options of the moment as such a word (ops.snet_algo_flags: a module's own dict on top)."""
    from transmf_ad_amd import _lib, ops
    lib = _lib.load()

    def saved(flags):
        desc = _lib.SnetDesc(B=8, D=96, H=96, W=96, dim=128, precision=0, storage_bf16=0, flags=flags)
        desc.momentum[:] = [0.1] * 7
        desc.eps[:] = [1e-5] * 7
        desc.slope[:] = [0.01] * 7
        return lib.tmf_snet_saved_bytes(ctypes.byref(desc))
    ALGO, P, X, G, S = 0x100, 0x800, 0x1000, 0x2000, 0x8000      # (S: c1_split, the first block's z as exact bf16 splits)
    assert lib.tmf_snet_algo_flags() == ALGO | (3 << 9) | P | X | G | S
    default = saved(0)
    assert saved(lib.tmf_snet_algo_flags()) == default
    by_flags = {m: saved(ALGO | (m << 9) | P | X | G | S) for m in (0, 1, 2, 3)}
    no_gram = saved(ALGO | (3 << 9) | P | X | S)
    assert by_flags[3] == default and len(set(by_flags.values())) >= 3 and no_gram < default
    assert saved(ALGO | (3 << 9) | P | X | G) == default                      # (c1_split changes kernels, not the plan)
    try:
        for m in (0, 1, 2):
            assert lib.tmf_set_option(b"conv_wino", m) == 0
            assert saved(0) == by_flags[m]                                    # the process option ...
            assert saved(ALGO | (3 << 9) | P | X | G | S) == default          # ... does not reach a call that carries its own
            assert lib.tmf_snet_algo_flags() == ALGO | (m << 9) | P | X | G | S
        assert lib.tmf_set_option(b"conv_wino", 3) == 0 and lib.tmf_set_option(b"c1_gram", 0) == 0
        assert saved(0) == no_gram and saved(ALGO | (3 << 9) | P | X | G | S) == default
        assert lib.tmf_set_option(b"c1_split", 0) == 0 and lib.tmf_c1_split_mode() == 0
        assert lib.tmf_snet_algo_flags() == ALGO | (3 << 9) | P | X
    finally:
        lib.tmf_set_option(b"conv_wino", 3)
        lib.tmf_set_option(b"c1_gram", 1)
        lib.tmf_set_option(b"c1_split", 1)
    assert ops.snet_algo_flags() == ALGO | (3 << 9) | P | X | G | S
    assert ops.snet_algo_flags(dict(conv_wino=0, wino_x=0)) == ALGO | P | G | S
    assert ops.snet_algo_flags(dict(c1_split=0, c1_gram=2)) == ALGO | (3 << 9) | P | X | G | 0x4000
    with pytest.raises(ValueError):
        ops.snet_algo_flags(dict(winograd=1))
    assert lib.tmf_conv_wino_mode() == 3 and lib.tmf_wino_x_mode() == 1 and lib.tmf_c1_split_mode() == 1   # (no override leaks out of a call)


def test_split_winograd_kernel_takes_the_launches_it_is_built_for():
    """tmf_wino_x_mode() / tmf_conv3d_wino_kernel_name2 (host logic of csrc/conv3d_winox.hip): the split kernel takes the train
    forward / data gradient where cin and cout are multiples of 32 and the volume uses 4x8x8 bricks of one sample; everything else —
    8- and 16-channel inputs, the folded four-sample geometry of the small deep volumes, wino_x 0 — stays on the fp32 kernels."""
    from transmf_ad_amd import _lib
    lib = _lib.load()
    assert lib.tmf_wino_x_mode() == 1
    name = lambda *a: lib.tmf_conv3d_wino_kernel_name2(*a)                   # noqa: E731
    assert name(8, 48, 48, 48, 32, 32, 1) == b"conv3d_winox_kernel<1>" and name(8, 48, 48, 48, 64, 32, 0) == b"conv3d_winox_kernel<0>"
    assert name(8, 24, 24, 24, 64, 128, 1) == b"conv3d_winox_kernel<1>"
    assert name(8, 48, 48, 48, 16, 32, 1) == b"conv3d_wino_p_kernel<1, 0>"   # cin % 32
    assert name(8, 12, 12, 12, 128, 256, 1) == b"conv3d_wino_p_kernel<1, 1>"  # four samples x 4x4x4 bricks
    # the geometry is chosen WITH the channel counts (round 6): 22x27x22 at B = 8 — the reference's 91x109x91 two levels down — is
    # 576 (504 transposed) one-sample bricks against 504 folded ones; the split kernel beats the fp32 kernel on that many, so it keeps
    # the launch where it can take it, and only there (the folded geometry must save more than 15 %)
    # (... and the split kernel lays its items with the 4-voxel side along h there: 7 x 3 x 3 = 63 per sample instead of 6 x 4 x 3 = 72)
    assert lib.tmf_conv3d_wino_bricks(8, 22, 27, 22) == 504 and lib.tmf_conv3d_wino_bricks2(8, 22, 27, 22, 64, 128) == 8 * 63
    assert name(8, 22, 27, 22, 64, 128, 1) == b"conv3d_winox_kernel<1>" and name(8, 22, 27, 22, 128, 64, 0) == b"conv3d_winox_kernel<0>"
    assert lib.tmf_conv3d_wino_bricks2(8, 22, 27, 22, 16, 64) == 504 and name(8, 22, 27, 22, 16, 64, 1) == b"conv3d_wino_p_kernel<1, 1>"
    assert lib.tmf_conv3d_wino_bricks2(8, 11, 13, 11, 128, 256) == 72 and name(8, 11, 13, 11, 128, 256, 1) == b"conv3d_wino_p_kernel<1, 1>"
    assert lib.tmf_conv3d_wino_bricks2(8, 48, 48, 48, 32, 64) == lib.tmf_conv3d_wino_bricks(8, 48, 48, 48)
    assert lib.tmf_set_option(b"wino_x", 0) == 0
    try:
        assert lib.tmf_wino_x_mode() == 0 and name(8, 48, 48, 48, 32, 32, 1) == b"conv3d_wino_p_kernel<1, 0>"
        assert lib.tmf_conv3d_wino_bricks2(8, 22, 27, 22, 64, 128) == 504 and name(8, 22, 27, 22, 64, 128, 1) == b"conv3d_wino_p_kernel<1, 1>"
    finally:
        lib.tmf_set_option(b"wino_x", 1)
    assert name(8, 48, 48, 48, 32, 32, 1) == b"conv3d_winox_kernel<1>"


def test_built_kernels_have_no_use_of_registers_in_flight_and_no_store_data_overwrite(tmp_path):
    """tools/asm_checks.py on the DISASSEMBLY of every built code object (tools/resources.disassembly_of: what ships, not a
    recompilation): no wide store's data registers are overwritten in the slot behind it (the MI355X stores a wrong value there;
    LLVM exempts the SGPR-soffset form — found in round 5 in the Winograd forward), and the weight loads the persistent Winograd
    kernels keep in flight to registers (inline asm) are not touched before their wait."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import asm_checks
    import resources as R
    if not os.path.exists(f"{R.LLVM}/llvm-objdump") or not os.path.exists(os.path.join(R.CSRC, "conv3d_wino.o")):
        pytest.skip("objects or ROCm llvm tools not present")
    seen = 0
    for f in sorted(os.listdir(R.CSRC)):
        if not f.endswith(".o"):
            continue
        out = str(tmp_path / (f + ".s"))
        if not R.disassembly_of(os.path.join(R.CSRC, f), out):
            continue                                                        # (host-only object)
        stores, overwrites = asm_checks.store_data_overwrites(out, "")
        assert not overwrites, (f, overwrites[:5])
        seen += stores
        if f == "conv3d_wino.o":
            loads, uses = asm_checks.inflight_uses(out, "wino_p_kernel")
            loads2, uses2 = asm_checks.inflight_uses(out, "wino_wgrad_p_kernel")
            assert loads >= 100 and stores >= 40                            # (the check has seen the kernels)
            assert not uses and not uses2, (uses + uses2)[:5]
        if f == "conv3d_winox.o":
            # counted waits (s_waitcnt vmcnt(n), loads / LDS-DMA copies / stores retire in order): a model of the queue walks every
            # control-flow edge of the kernels and finds no access to a register whose load is still outstanding
            import asm_inflight
            res = asm_inflight.check_object(os.path.join(R.CSRC, f), "winox_kernel")
            assert len(res) == 4                                              # train forward, data gradient, the two eval-mode blocks
            for name, bad, loads, depth in res:
                assert loads >= 40 and depth <= 63 and not bad, (name, bad[:5])
    assert seen >= 300


def test_first_block_gram_path_is_offered_only_inside_its_limits():
    """tmf_c1_gram_bytes (csrc/conv1_gram.hip): 0 — the callers then take the recomputing passes — beyond 64 channels, beyond the
    32-bit voxel offsets / 10-bit coordinates of its kernels, and with the option off; the backward workspace covers the sums, the
    D slabs, their reduction scratch and the reduced D."""
    from transmf_ad_amd import _lib
    lib = _lib.load()
    n = lib.tmf_c1_gram_bytes(8, 96, 96, 96, 32)
    assert n == (760 + 256 * 64 + 6 * 64 * 96) * 8
    assert lib.tmf_c1_gram_bytes(8, 96, 96, 96, 128) == 0 and lib.tmf_c1_gram_bytes(1, 1100, 64, 64, 32) == 0
    assert lib.tmf_c1_gram_bytes(64, 400, 400, 400, 32) == 0 and lib.tmf_c1_gram_bytes(0, 96, 96, 96, 32) == 0
    lib.tmf_c1_gram_bytes_bf16.restype = ctypes.c_size_t
    assert lib.tmf_c1_gram_bytes_bf16(8, 96, 96, 96, 32) == 0           # the bf16 mode: under "c1_gram" 2 only
    assert lib.tmf_set_option(b"c1_gram", 2) == 0
    try:
        assert lib.tmf_c1_gram_bytes_bf16(8, 96, 96, 96, 32) == n and lib.tmf_c1_gram_bytes(8, 96, 96, 96, 32) == n
        assert lib.tmf_snet_algo_flags() & 0x6000 == 0x6000
    finally:
        lib.tmf_set_option(b"c1_gram", 1)
    assert lib.tmf_snet_algo_flags() & 0x6000 == 0x2000
    assert lib.tmf_set_option(b"c1_gram", 0) == 0
    try:
        assert lib.tmf_c1_gram_bytes(8, 96, 96, 96, 32) == 0
    finally:
        lib.tmf_set_option(b"c1_gram", 1)
    nblk = lib.tmf_c1_blocks(8, 96, 96, 96, 32)
    assert lib.tmf_c1_bwd_fused_workspace_bytes(8, 96, 96, 96, 32) >= (nblk * 2 * 32 + (nblk + 1) * 27 * 32) * 4


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` without a torchrun environment composes the driver's launch line and starts it as a CHILD
    process before any GPU call; with fewer GPUs than ranks it says so."""
    import subprocess
    import sys
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.launch_command(["--gpus", "4", "--steps", "7", "--warmup", "2"], 4, 29611, python="python3")
    assert cmd[:3] == ["python3", "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29611"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]          # the same arguments, after the script
    assert bench.visible_gpu_error(2, 8) is None and bench.visible_gpu_error(8, 8) is None
    assert "2 GPUs requested, 1 visible" in bench.visible_gpu_error(2, 1)
    # end to end on this box (no GPU): the self-launch path is taken and refuses before any child is started
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "2 GPUs requested, 0 visible" in r.stderr and "launch with torch.distributed.run" not in r.stderr
    # inside a launcher's environment a mismatch is still an error of its own
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env2, capture_output=True,
                        text=True, timeout=300)
    assert r2.returncode != 0 and "--nproc-per-node must equal --gpus" in r2.stderr
