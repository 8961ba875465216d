"""Input pipeline ahead of the path: the numpy restatement of MONAI's ScaleIntensity / Flip
(oracle/input_oracle.py) on the host, and — on the GPU — the HIP kernels bit-for-bit against it, through the C ABI and
through the double-buffered DevicePrefetcher."""
import numpy as np
import pytest
import torch

from oracle import input_oracle as IO

DEV = "cuda:0"


def _raw(B, shape, seed, lo=-50.0, hi=4000.0):
    rs = np.random.RandomState(seed)
    return (rs.rand(B, 1, *shape) * (hi - lo) + lo).astype(np.float32)


def test_oracle_scale_intensity_properties():
    v = _raw(1, (9, 10, 11), 0)[0]
    s = IO.scale_intensity(v)
    assert s.dtype == np.float32 and s.min() == 0.0 and s.max() == 1.0
    assert np.array_equal(IO.scale_intensity(s), s)                      # idempotent on [0, 1] data with min 0 / max 1
    assert np.array_equal(IO.scale_intensity(np.full((1, 3, 3, 3), 7.0, np.float32)), np.zeros((1, 3, 3, 3), np.float32))
    f = IO.rand_flip(v, True)
    assert np.array_equal(f[:, 0], v[:, -1]) and np.array_equal(IO.rand_flip(f, True), v)     # involution, axis 1 of (C, D, H, W)
    assert IO.rand_flip(v, False) is v


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(16, 16, 16), (91, 109, 91), (7, 5, 3), (32, 20, 6)])
def test_scale_flip_kernels_bit_exact(shape):
    import transmf_ad_amd as T
    B = 5
    mri, pet = _raw(B, shape, 1), _raw(B, shape, 2, lo=0.0, hi=1.0)
    mri[3] = 12.5                                            # a constant volume -> zeros
    flips = np.array([1, 0, 1, 1, 0], np.uint8)
    want_m, want_p = IO.train_transform(mri, pet, flips)
    fl = torch.from_numpy(flips).to(DEV)
    got_m = T.scale_intensity_flip(torch.from_numpy(mri).to(DEV), fl)
    got_p = T.scale_intensity_flip(torch.from_numpy(pet).to(DEV), fl)
    torch.cuda.synchronize()
    assert np.array_equal(got_m.cpu().numpy(), want_m)        # bit for bit
    assert np.array_equal(got_p.cpu().numpy(), want_p)
    no_flip = T.scale_intensity_flip(torch.from_numpy(mri).to(DEV), None).cpu().numpy()
    assert np.array_equal(no_flip, IO.train_transform(mri, pet, np.zeros(B, np.uint8))[0])


@pytest.mark.gpu
def test_scale_intensity_propagates_nan_like_numpy():
    import transmf_ad_amd as T
    v = _raw(2, (8, 8, 8), 3)
    v[1, 0, 2, 3, 4] = np.nan
    got = T.scale_intensity_flip(torch.from_numpy(v).to(DEV)).cpu().numpy()
    assert np.array_equal(got[0], IO.scale_intensity(v[0]))
    assert np.isnan(got[1]).all() and np.isnan(IO.scale_intensity(v[1])).all()


@pytest.mark.gpu
def test_device_prefetcher_yields_reference_transform():
    """Three host batches through the pinned, double-buffered prefetcher: every device batch equals the oracle transform
    of its host batch with the flip decisions the prefetcher drew, labels intact, order preserved."""
    import transmf_ad_amd as T
    shape = (24, 20, 16)
    host = [dict(MRI=_raw(4, shape, 10 + i), PET=_raw(4, shape, 20 + i), label=np.arange(4) % 2 + 0 * i) for i in range(3)]
    seen = 0
    for i, batch in enumerate(T.DevicePrefetcher(host, device=DEV, flip_prob=0.5, seed=7)):
        wm, wp = IO.train_transform(host[i]["MRI"], host[i]["PET"], batch["_flips"])
        torch.cuda.synchronize()
        assert np.array_equal(batch["MRI"].cpu().numpy(), wm) and np.array_equal(batch["PET"].cpu().numpy(), wp)
        assert batch["label"].dtype == torch.int64 and batch["label"].tolist() == list(host[i]["label"])
        seen += 1
    assert seen == 3
    # evaluation: no flips
    for i, batch in enumerate(T.DevicePrefetcher(host[:1], device=DEV, train=False)):
        assert not batch["_flips"].any()
    with pytest.raises(NotImplementedError):
        T.DevicePrefetcher(host, device=DEV, strict_reference_aug=True)
