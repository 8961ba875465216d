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
    for i, batch in enumerate(T.DevicePrefetcher(host, device=DEV, flip_prob=0.5, seed=7, rotate_prob=0.0, zoom_prob=0.0)):
        wm, wp = IO.train_transform(host[i]["MRI"], host[i]["PET"], batch["_flips"])
        torch.cuda.synchronize()
        assert np.array_equal(batch["MRI"].cpu().numpy(), wm) and np.array_equal(batch["PET"].cpu().numpy(), wp)
        assert batch["label"].dtype == torch.int64 and batch["label"].tolist() == list(host[i]["label"])
        seen += 1
    assert seen == 3
    # evaluation: no augmentation at all
    for i, batch in enumerate(T.DevicePrefetcher(host[:1], device=DEV, train=False)):
        assert not batch["_flips"].any() and np.isnan(batch["_angles"]).all() and np.isnan(batch["_zooms"]).all()


def test_oracle_zoom_is_torch_area_interpolation_bit_for_bit():
    """`zoom_area` = F.interpolate(mode="area") to floor(S z) + edge (replicate) padding back to S — MONAI's Zoom(mode="area",
    padding_mode="edge", keep_size=True) for z <= 1 — bit for bit on the host."""
    import torch.nn.functional as F
    rs = np.random.RandomState(4)
    for shape in [(12, 14, 10), (24, 24, 24), (31, 37, 29)]:
        v = rs.rand(1, *shape).astype(np.float32)
        for z in (0.95, 0.9712, 0.999, 1.0):
            So = IO.zoom_out_size(shape, z)
            t = F.interpolate(torch.from_numpy(v)[None], size=So, mode="area")
            pads = []
            for S, O in zip(shape[::-1], So[::-1]):
                pads += [(S - O) // 2, S - O - (S - O) // 2]
            want = F.pad(t, pads, mode="replicate")[0].numpy()
            assert np.array_equal(IO.zoom_area(v, z), want), (shape, z)
    with pytest.raises(ValueError):
        IO.zoom_area(np.zeros((1, 4, 4, 4), np.float32), 1.3)


def _monai_rotate_route(v, ang):
    """MONAI's Rotate on the host, spelled out with torch: index-space pull transform T(c) Rx T(-c), to_norm_affine
    (align_corners=False), reverse indexing, F.affine_grid + F.grid_sample(bilinear, border)."""
    import math
    import torch.nn.functional as F
    shape = v.shape[1:]
    c, s_ = math.cos(ang), math.sin(ang)
    R = np.eye(4); R[1, 1] = c; R[1, 2] = -s_; R[2, 1] = s_; R[2, 2] = c
    ctr = [(k - 1) / 2 for k in shape]
    T1 = np.eye(4); T1[:3, 3] = ctr
    T2 = np.eye(4); T2[:3, 3] = [-k for k in ctr]
    A = T1 @ R @ T2
    Nn = np.eye(4)
    for i, S in enumerate(shape):
        Nn[i, i] = 2.0 / S
        Nn[i, 3] = 1.0 / S - 1.0
    th = Nn @ A @ np.linalg.inv(Nn)
    th[:3] = th[[2, 1, 0]]
    th[:, :3] = th[:, [2, 1, 0]]
    grid = F.affine_grid(torch.tensor(th[:3], dtype=torch.float32)[None], [1, 1, *shape], align_corners=False)
    return F.grid_sample(torch.from_numpy(v)[None], grid, mode="bilinear", padding_mode="border", align_corners=False)[0].numpy()


def test_oracle_rotate_follows_the_grid_sample_route():
    """`rotate_x` (in-plane bilinear pull rotation, fixed fp32 operation order) against MONAI's route through normalised
    coordinates and torch's grid_sample: the two differ only by the ~1e-6-voxel perturbation of the sampling point that the
    normalisation round trip introduces (times the local gradient: uniform-noise volumes are the worst case)."""
    rs = np.random.RandomState(5)
    for shape in [(12, 14, 10), (24, 24, 24), (45, 54, 45)]:
        v = rs.rand(1, *shape).astype(np.float32)
        for ang in (0.05, -0.031, 0.0):
            assert np.abs(IO.rotate_x(v, ang) - _monai_rotate_route(v, ang)).max() < 2e-5, (shape, ang)
        assert np.array_equal(IO.rotate_x(v, 0.0), v)            # angle 0 is the identity, exactly
    # a rotation moves content: the centre stays, an off-centre blob moves by ~angle * radius
    v = np.zeros((1, 3, 41, 41), np.float32); v[:, :, 20, 35] = 1.0
    r = IO.rotate_x(v, 0.05)
    assert abs(r[0, 1].sum() - 1.0) < 1e-3 and r[0, 1, 20, 35] < 0.9 and r[0, 1, 20, 20] == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(16, 16, 16), (91, 109, 91), (7, 5, 3), (24, 20, 18)])
def test_rotate_zoom_kernels_bit_exact(shape):
    """tmf_rotate_x / tmf_zoom_area against the numpy restatement, bit for bit: mixed decisions in one batch (rotate only,
    zoom only, both, neither), both signs of the angle, the extreme zoom factors."""
    import transmf_ad_amd as T
    B = 6
    x = _raw(B, shape, 31, lo=0.0, hi=1.0)
    angles = np.array([0.05, np.nan, -0.0312, np.nan, 0.0123, -0.05])
    zooms = np.array([np.nan, 0.95, 0.9731, np.nan, 0.9999, 0.951])
    want = np.stack([x[b] for b in range(B)])
    for b in range(B):
        y = x[b]
        if not np.isnan(angles[b]):
            y = IO.rotate_x(y, angles[b])
        if not np.isnan(zooms[b]):
            y = IO.zoom_area(y, zooms[b])
        want[b] = y
    got = T.rotate_zoom(torch.from_numpy(x).to(DEV), angles, zooms)
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy(), want)
    assert torch.equal(T.rotate_zoom(torch.from_numpy(x).to(DEV), None, None).cpu(), torch.from_numpy(x))


@pytest.mark.gpu
def test_device_prefetcher_full_reference_augmentation():
    """The default DevicePrefetcher applies the reference's whole train transform (datasets/ADNI.py:64-68): every device
    batch equals ScaleIntensity -> flip -> rotate -> zoom of its host batch in the numpy restatement, with the decisions the
    prefetcher drew (shared by MRI and PET); with probability 1 every subject gets all three."""
    import transmf_ad_amd as T
    shape = (20, 24, 16)
    host = [dict(MRI=_raw(4, shape, 40 + i), PET=_raw(4, shape, 50 + i), label=np.arange(4) % 2) for i in range(3)]
    n_rot = n_zoom = 0
    for i, batch in enumerate(T.DevicePrefetcher(host, device=DEV, seed=11, rotate_prob=0.6, zoom_prob=0.6)):
        wm, wp = IO.train_transform(host[i]["MRI"], host[i]["PET"], batch["_flips"], batch["_angles"], batch["_zooms"])
        torch.cuda.synchronize()
        assert np.array_equal(batch["MRI"].cpu().numpy(), wm) and np.array_equal(batch["PET"].cpu().numpy(), wp)
        a, z = batch["_angles"], batch["_zooms"]
        assert (np.abs(a[~np.isnan(a)]) <= 0.05).all() and ((z[~np.isnan(z)] >= 0.95) & (z[~np.isnan(z)] < 1.0)).all()
        n_rot += (~np.isnan(a)).sum()
        n_zoom += (~np.isnan(z)).sum()
    assert 0 < n_rot < 12 and 0 < n_zoom < 12
    for batch in T.DevicePrefetcher(host[:1], device=DEV, seed=1, flip_prob=1.0, rotate_prob=1.0, zoom_prob=1.0):
        assert batch["_flips"].all() and not np.isnan(batch["_angles"]).any() and not np.isnan(batch["_zooms"]).any()


def test_nifti_reader_round_trip_and_header_rules(tmp_path):
    """read_nifti against files written by write_nifti and by hand: float32 / int16 / uint8 data, first index fastest,
    scl_slope / scl_inter applied (slope 0 and NaN = no scaling), gzip, both byte orders; malformed files raise."""
    from transmf_ad_amd import nifti as NI
    rs = np.random.RandomState(8)
    a = (rs.rand(5, 7, 3) * 100).astype(np.float32)
    for name, be in (("a.nii", False), ("a.nii.gz", False), ("b.nii", True)):
        NI.write_nifti(str(tmp_path / name), a, big_endian=be)
        got = NI.read_nifti(str(tmp_path / name))
        assert got.dtype == np.float32 and got.shape == (5, 7, 3) and np.array_equal(got, a)
    i16 = (rs.rand(4, 6, 5) * 2000 - 1000).astype(np.int16)
    NI.write_nifti(str(tmp_path / "i.nii.gz"), i16, slope=0.5, inter=-3.0)
    assert np.array_equal(NI.read_nifti(str(tmp_path / "i.nii.gz")), (i16.astype(np.float64) * 0.5 - 3.0).astype(np.float32))
    NI.write_nifti(str(tmp_path / "n.nii"), i16, slope=float("nan"), inter=9.0)
    assert np.array_equal(NI.read_nifti(str(tmp_path / "n.nii")), i16.astype(np.float32))      # NaN slope: unscaled
    NI.write_nifti(str(tmp_path / "inf.nii"), i16, slope=float("inf"), inter=9.0)
    assert np.array_equal(NI.read_nifti(str(tmp_path / "inf.nii")), i16.astype(np.float32))    # non-finite slope: unscaled
    NI.write_nifti(str(tmp_path / "badinter.nii"), i16, slope=2.0, inter=float("inf"))
    with pytest.raises(NI.NiftiError):
        NI.read_nifti(str(tmp_path / "badinter.nii"))                                          # valid slope, invalid intercept
    # a 4-D file with a trailing singleton axis (dim[0] = 4, dim[4] = 1: common for ADNI exports) is ONE 3-D volume for
    # nifti_batches, as for MONAI's LoadImaged + EnsureChannelFirstd; a real 4-D series is refused
    NI.write_nifti(str(tmp_path / "t1.nii"), i16[..., None])
    assert NI.read_nifti(str(tmp_path / "t1.nii")).shape == i16.shape + (1,)
    b4 = next(NI.nifti_batches([str(tmp_path / "t1.nii")], [str(tmp_path / "t1.nii")], [0], 1))
    assert b4["MRI"].shape == (1, 1) + i16.shape and np.array_equal(b4["MRI"][0, 0], i16.astype(np.float32))
    NI.write_nifti(str(tmp_path / "t2.nii"), np.stack([i16, i16], axis=-1))
    with pytest.raises(NI.NiftiError):
        next(NI.nifti_batches([str(tmp_path / "t2.nii")], [str(tmp_path / "t2.nii")], [0], 1))
    u8 = rs.randint(0, 255, (3, 2, 4)).astype(np.uint8)
    NI.write_nifti(str(tmp_path / "u.nii"), u8)
    assert np.array_equal(NI.read_nifti(str(tmp_path / "u.nii")), u8.astype(np.float32))
    # a header assembled by hand: the element order in the file is x fastest
    import struct
    hdr = bytearray(352)
    struct.pack_into("<i", hdr, 0, 348)
    struct.pack_into("<8h", hdr, 40, 3, 2, 3, 2, 1, 1, 1, 1)
    struct.pack_into("<2h", hdr, 70, 16, 32)
    struct.pack_into("<3f", hdr, 108, 352.0, 0.0, 0.0)
    hdr[344:348] = b"n+1\x00"
    vals = np.arange(12, dtype="<f4")
    (tmp_path / "h.nii").write_bytes(bytes(hdr) + vals.tobytes())
    h = NI.read_nifti(str(tmp_path / "h.nii"))
    assert h.shape == (2, 3, 2) and h[1, 0, 0] == 1.0 and h[0, 1, 0] == 2.0 and h[0, 0, 1] == 6.0
    for bad in (b"short", bytes(352)):
        (tmp_path / "bad.nii").write_bytes(bad)
        with pytest.raises(NI.NiftiError):
            NI.read_nifti(str(tmp_path / "bad.nii"))
    # batches for the prefetcher
    paths = []
    for k in range(5):
        NI.write_nifti(str(tmp_path / f"m{k}.nii.gz"), a + k)
        NI.write_nifti(str(tmp_path / f"p{k}.nii.gz"), a - k)
        paths.append((str(tmp_path / f"m{k}.nii.gz"), str(tmp_path / f"p{k}.nii.gz")))
    bs = list(NI.nifti_batches([m for m, _ in paths], [p_ for _, p_ in paths], [0, 1, 0, 1, 1], batch_size=2))
    assert [b["MRI"].shape for b in bs] == [(2, 1, 5, 7, 3), (2, 1, 5, 7, 3), (1, 1, 5, 7, 3)]
    assert np.array_equal(bs[1]["PET"][1, 0], a - 3) and bs[2]["label"].tolist() == [1]
    assert len(list(NI.nifti_batches([m for m, _ in paths], [p_ for _, p_ in paths], [0, 1, 0, 1, 1], 2, drop_last=True))) == 2


@pytest.mark.gpu
def test_nifti_files_through_the_prefetcher(tmp_path):
    """.nii.gz pairs -> nifti_batches -> DevicePrefetcher (read in its worker thread) -> device batches equal to the oracle
    transform of the file contents."""
    import transmf_ad_amd as T
    rs = np.random.RandomState(9)
    vols = [(rs.rand(16, 20, 12) * 900).astype(np.float32) for _ in range(8)]
    mp, pp = [], []
    for k in range(4):
        T.write_nifti(str(tmp_path / f"m{k}.nii.gz"), vols[2 * k])
        T.write_nifti(str(tmp_path / f"p{k}.nii.gz"), vols[2 * k + 1])
        mp.append(str(tmp_path / f"m{k}.nii.gz")); pp.append(str(tmp_path / f"p{k}.nii.gz"))
    n = 0
    for i, batch in enumerate(T.DevicePrefetcher(T.nifti_batches(mp, pp, [0, 1, 1, 0], 2), device=DEV, seed=3)):
        hm = np.stack([vols[2 * k][None] for k in (2 * i, 2 * i + 1)])
        hp = np.stack([vols[2 * k + 1][None] for k in (2 * i, 2 * i + 1)])
        wm, wp = IO.train_transform(hm, hp, batch["_flips"], batch["_angles"], batch["_zooms"])
        torch.cuda.synchronize()
        assert np.array_equal(batch["MRI"].cpu().numpy(), wm) and np.array_equal(batch["PET"].cpu().numpy(), wp)
        n += 1
    assert n == 2
