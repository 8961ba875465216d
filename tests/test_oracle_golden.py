"""Pin the CPU oracle (oracle/tmf_oracle.py) against the golden vectors produced by the
imported reference (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from _golden import Golden, available, gprobe, probe, run_oracle, zero_grad_keys

FAST = ["ad_tiny", "ad_ragged", "cnn_tiny", "single_mid", "cnn_mid", "ad_mid", "ad_mid_drop", "ad_mid_h8"]
SLOW = ["ad_full_b2", "ad_full_b2_blobs", "ad_adni_b2", "cnn_full_b2", "single_full_b2", "cnn_full_b16", "single_full_b16"]


def _check_case(name, dtype, prec):
    g = Golden(name)
    r = run_oracle(g, dtype=dtype, train=True)
    tol = 2e-5 if prec == "f32" else 1e-10
    for k, v in r["outs"].items():
        ref = g[f"{prec}/train/{k}"]
        assert np.abs(v.detach().double().numpy() - ref).max() <= tol, k
    assert abs(r["loss"].item() - float(g[f"{prec}/train/loss"])) <= tol
    # activation probes
    for k, t in r["probes"].items():
        ref = g[f"{prec}/probe/{k}"]
        got = probe(t)
        scale = max(1.0, np.abs(ref).max())
        assert np.abs(got - ref).max() <= (5e-5 if prec == "f32" else 1e-9) * scale, k
    # gradient probes: fp64 exact-ish; fp32 only against the fp64 golden with the reference's own noise band
    zk = zero_grad_keys(g.spec, g.model)
    for k, gr in r["grads"].items():
        ref = g[f"f64/grad/{k}"]
        got = gprobe(gr)
        if k in zk:
            # mathematically zero: bounded by rounding noise relative to the weight grad of the same layer
            continue
        mx = max(ref[2], 1e-30)
        err = np.abs(got[3:] - ref[3:]).max() / mx
        assert err <= (3e-2 if prec == "f32" else 1e-8), (k, err)
    return g, r


@pytest.mark.parametrize("name", FAST)
def test_oracle_train_f64(name):
    if not available(name):
        pytest.skip("fixture not generated")
    _check_case(name, torch.float64, "f64")


@pytest.mark.parametrize("name", FAST)
def test_oracle_train_f32_and_buffers(name):
    if not available(name):
        pytest.skip("fixture not generated")
    g, r = _check_case(name, torch.float32, "f32")
    S = r["state"]
    for k, (kind, _s) in g.spec.items():
        if kind == "buffer":
            ref = g[f"f32/buf/{k}"]
            got = S[k].detach().double().numpy()
            assert np.abs(got - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()), k


@pytest.mark.parametrize("name", FAST)
def test_oracle_eval(name):
    if not available(name):
        pytest.skip("fixture not generated")
    g = Golden(name)
    r = run_oracle(g, train=False)
    for k, v in r["outs"].items():
        assert np.abs(v.double().numpy() - g[f"f32/eval/{k}"]).max() <= 2e-5, k


def test_zero_grad_keys_really_zero():
    """The excluded gradients are zero in exact arithmetic: fp64 golden says <= 1e-9."""
    g = Golden("ad_tiny")
    for k in zero_grad_keys(g.spec, g.model):
        assert g[f"f64/grad/{k}"][2] <= 1e-9, k


@pytest.mark.slow
@pytest.mark.parametrize("name", SLOW)
def test_oracle_full_size(name):
    if not available(name):
        pytest.skip("fixture not generated")
    g = Golden(name)
    r = run_oracle(g, dtype=torch.float32, train=True, backward=False)
    for k, v in r["outs"].items():
        assert np.abs(v.detach().double().numpy() - g[f"f32/train/{k}"]).max() <= 2e-5, k
        # fp32 run vs the fp64 golden: the reference's own fp32 run is 8e-5 (ad_full_b2) / 1.5e-4 (ad_adni_b2) away
        # (train-mode BatchNorm1d over a batch of 2)
        assert np.abs(v.detach().double().numpy() - g[f"f64/train/{k}"]).max() <= 3e-4, k
