"""Cross-check of the two oracle restatements: the plain-C loops (oracle/tmf_oracle.c, fp64 accumulation)
against the torch-functional oracle's ops, on a tiny sNet-shaped chain and a tiny attention.  CPU only."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ORC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")


@pytest.fixture(scope="module")
def lib():
    subprocess.check_call(["make", "-s", "-C", ORC])
    return C.CDLL(os.path.join(ORC, "libtmf_oracle.so"))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_conv_bn_lrelu_pool_chain(lib):
    rs = np.random.RandomState(0)
    B, Cin, Cout, D, H, W = 2, 3, 5, 6, 5, 7
    x = rs.rand(B, Cin, D, H, W).astype(np.float32)
    w = (rs.standard_normal((Cout, Cin, 3, 3, 3)) * 0.2).astype(np.float32)
    b = (rs.standard_normal(Cout) * 0.1).astype(np.float32)
    g = (1 + 0.1 * rs.standard_normal(Cout)).astype(np.float32)
    be = (0.1 * rs.standard_normal(Cout)).astype(np.float32)
    rm, rv = np.zeros(Cout, np.float32), np.ones(Cout, np.float32)
    y = np.empty((B, Cout, D, H, W), np.float32)
    lib.orc_conv3d(_p(x), _p(w), _p(b), _p(y), B, Cin, Cout, D, H, W, 3)
    lib.orc_batchnorm_train(_p(y), _p(g), _p(be), _p(rm), _p(rv), B, Cout, C.c_long(D * H * W),
                            C.c_float(0.1), C.c_float(1e-5))
    lib.orc_leaky_relu(_p(y), C.c_long(y.size), C.c_float(0.01))
    for mode, fn in ((1, F.max_pool3d), (2, F.avg_pool3d)):
        out = np.empty((B, Cout, D // 2, H // 2, W // 2), np.float32)
        lib.orc_pool2(_p(y), _p(out), B * Cout, D, H, W, mode)
        trm, trv = torch.zeros(Cout), torch.ones(Cout)
        t = F.conv3d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), padding=1)
        t = F.leaky_relu(F.batch_norm(t, trm, trv, torch.from_numpy(g), torch.from_numpy(be), True, 0.1, 1e-5), 0.01)
        assert np.abs(fn(t, 2, 2).numpy() - out).max() < 2e-5
        assert np.abs(trm.numpy() - rm).max() < 1e-6 and np.abs(trv.numpy() - rv).max() < 1e-6


def test_conv1x1(lib):
    rs = np.random.RandomState(1)
    x = rs.standard_normal((1, 6, 3, 2, 4)).astype(np.float32)
    w = rs.standard_normal((4, 6, 1, 1, 1)).astype(np.float32)
    y = np.empty((1, 4, 3, 2, 4), np.float32)
    lib.orc_conv3d(_p(x), _p(w), None, _p(y), 1, 6, 4, 3, 2, 4, 1)
    assert np.abs(F.conv3d(torch.from_numpy(x), torch.from_numpy(w)).numpy() - y).max() < 1e-5


def test_transformer_pieces(lib):
    from oracle import tmf_oracle as O
    rs = np.random.RandomState(2)
    B, N, M, heads, dh, dim = 2, 5, 7, 2, 4, 8
    inner = heads * dh
    x = rs.standard_normal((B, N, dim)).astype(np.float32)
    ctx = rs.standard_normal((B, M, dim)).astype(np.float32)
    S = {"fn.to_q.weight": torch.from_numpy(rs.standard_normal((inner, dim)).astype(np.float32) * 0.3),
         "fn.to_kv.weight": torch.from_numpy(rs.standard_normal((2 * inner, dim)).astype(np.float32) * 0.3),
         "fn.to_out.0.weight": torch.from_numpy(rs.standard_normal((dim, inner)).astype(np.float32) * 0.3),
         "fn.to_out.0.bias": torch.from_numpy(rs.standard_normal(dim).astype(np.float32) * 0.1)}
    ref = O.attention_forward(S, "fn.", torch.from_numpy(x), torch.from_numpy(ctx), heads).numpy()
    q = np.empty((B, N, inner), np.float32)
    kv = np.empty((B, M, 2 * inner), np.float32)
    lib.orc_linear(_p(x), _p(S["fn.to_q.weight"].numpy()), None, _p(q), C.c_long(B * N), dim, inner)
    lib.orc_linear(_p(ctx), _p(S["fn.to_kv.weight"].numpy()), None, _p(kv), C.c_long(B * M), dim, 2 * inner)
    k = np.ascontiguousarray(kv[..., :inner])
    v = np.ascontiguousarray(kv[..., inner:])
    att = np.empty((B, N, inner), np.float32)
    scratch = np.empty(M, np.float64)
    lib.orc_attention(_p(q), _p(k), _p(v), _p(att), B, heads, N, M, dh, C.c_float(dh ** -0.5), _p(scratch))
    out = np.empty((B, N, dim), np.float32)
    lib.orc_linear(_p(att), _p(S["fn.to_out.0.weight"].numpy()), _p(S["fn.to_out.0.bias"].numpy()), _p(out),
                   C.c_long(B * N), inner, dim)
    assert np.abs(out - ref).max() < 2e-6
    # LayerNorm + GELU
    g = (1 + 0.1 * rs.standard_normal(dim)).astype(np.float32)
    b = (0.1 * rs.standard_normal(dim)).astype(np.float32)
    ln = np.empty_like(x)
    lib.orc_layernorm(_p(x), _p(g), _p(b), _p(ln), C.c_long(B * N), dim, C.c_float(1e-5))
    assert np.abs(F.layer_norm(torch.from_numpy(x), (dim,), torch.from_numpy(g), torch.from_numpy(b)).numpy() - ln).max() < 2e-6
    ge = x.copy()
    lib.orc_gelu(_p(ge), C.c_long(ge.size))
    assert np.abs(F.gelu(torch.from_numpy(x)).numpy() - ge).max() < 1e-6
