"""Register / scratch budget of the product kernels, read from the built code objects (no GPU, no recompilation).

The two-workgroups-per-CU configurations of the convolution kernels only work below 128 registers per wave; a kernel
edit once pushed conv3d_fwd_kernel to 130-156 registers and cost 0.4 ms per step without any error.  This test makes
that class of regression fail loudly."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tools import resources as R       # noqa: E402


@pytest.fixture(scope="module")
def kernels():
    if not os.path.exists(os.path.join(R.CSRC, "conv3d_mfma.o")):
        pytest.skip("objects not built (python -m transmf_ad_amd.build)")
    if not os.path.exists(f"{R.LLVM}/clang-offload-bundler"):
        pytest.skip("ROCm llvm tools not present")
    return R.all_kernels()


def _find(kernels, obj, *needles):
    ks = [k for k in kernels[obj] if all(n in k["name"] for n in needles)]
    assert ks, (obj, needles)
    return ks


def test_two_workgroup_conv_kernels_fit_128_registers_without_scratch(kernels):
    for nt in (1, 2):       # FwdCfg<3, 16, 1, NT, 8, 1, 4, 8, 8, 3>, VEC = true, raw (Lb0) and fused-eval (Lb1) epilogues
        for k in _find(kernels, "conv3d_mfma.o", f"conv3d_fwd_kernelINS_6FwdCfgILi3ELi16ELi1ELi{nt}ELi8ELi1ELi4ELi8ELi8ELi3EEELb1E"):
            assert k["vgpr"] <= 128 and k.get("scratch", 0) == 0, k
    for k in _find(kernels, "conv3d_bf16.o", "conv3d_fwd_bf16_kernelILi1E"):
        assert k["vgpr"] <= 128 and k.get("scratch", 0) == 0, k


def test_hot_kernels_do_not_spill(kernels):
    checks = [("conv3d_mfma.o", "conv3d_wgrad_kernelINS_5WgCfgILi1ELi4ELi8ELi8ELi8ELi32EEELb1E"),
              ("conv3d_mfma.o", "conv3d_wgrad_kernelINS_5WgCfgILi1ELi4ELi4ELi4ELi8ELi32EEELb1E"),
              ("conv3d_mfma.o", "conv3d_wgrad_kernelINS_5WgCfgILi1ELi4ELi8ELi8ELi8ELi16EEELb1E"),
              ("conv3d_mfma.o", "conv3d_wgrad_kernelINS_5WgCfgILi1ELi4ELi4ELi4ELi8ELi16EEELb1E"),
              ("conv3d_bf16.o", "conv3d_wgrad_bf16_kernel"), ("conv3d_bf16.o", "conv3d_fwd_split_kernel"),
              ("conv1_fused.o", "conv1_fused_kernel"), ("bn_act_pool.o", "bn_"), ("attention.o", "xattn_"),
              ("token_gemm.o", "tok_"), ("token_ops.o", "layernorm_")]
    for obj, needle in checks:
        for k in _find(kernels, obj, needle):
            assert k.get("scratch", 0) == 0, k


def test_no_packed_fp32_half_is_overwritten_unread(tmp_path):
    """gfx950 hazard found in round 1: `v_pk_*_f32 v[d:d+1]` followed by a single-pass vector instruction
    that overwrites one half of the pair before anything has read it intermittently keeps the stale half in lanes 16-31.
    clang's SLP vectoriser produces that sequence; the library is built with -fno-slp-vectorize.  This test compiles every
    source to a listing with the build's flags and fails if the sequence is back."""
    import subprocess
    from transmf_ad_amd import build as B
    from tools import pk_waw_scan as S
    hipcc = B._hipcc()
    procs = []
    for src in B.SOURCES:
        out = tmp_path / (src.replace(".hip", ".s"))
        cmd = [hipcc, "-x", "hip", "-S", "--cuda-device-only", os.path.join(B.CSRC, src), "-o", str(out)] + \
              [f for f in B.FLAGS if f != "-fPIC"] + ["-I" + os.path.join(B.HERE, "..", "include")]
        procs.append((src, out, subprocess.Popen(cmd, stderr=subprocess.DEVNULL)))
    bad = {}
    for src, out, p in procs:
        assert p.wait() == 0, src
        hits = S.scan(str(out))
        if hits:
            bad[src] = {k[:80]: v[:2] for k, v in list(hits.items())[:3]}
    assert not bad, bad
