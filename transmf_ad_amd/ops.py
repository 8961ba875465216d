"""torch.autograd.Functions over the C ABI of libtmf_hip.so.

PyTorch is plumbing here (device memory, the current HIP stream, autograd graph);
every numeric step of the hot path is a hand-written gfx950 kernel.  All tensors are
fp32, contiguous, on a HIP device; activations are channels-last (B, D, H, W, C).
"""
from __future__ import annotations

import os

import torch

from . import _lib

_f32 = torch.float32


def _stream():
    """Raw hipStream_t of torch's current stream (the private fast path: the public current_stream() builds a
    Stream object and costs ~10 us, ~0.7 ms per train step over ~75 calls)."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def _chk(t: torch.Tensor, name: str, allow_bf16: bool = False) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.TmfError(
            f"{name} is on {t.device}: transmf_ad_amd runs only on a HIP device (MI355X); there is no CPU fallback")
    if t.device.index != torch._C._cuda_getDevice():
        # the kernels are launched on the CURRENT device's current stream (_stream): pointers of another GPU there
        # would fault or race.  The module forwards install the guard themselves (networks.on_device_of).
        raise _lib.TmfError(
            f"{name} is on {t.device} but the current HIP device is cuda:{torch._C._cuda_getDevice()}: "
            f"call under `with torch.cuda.device({t.device.index}):`")
    if t.dtype != _f32 and not (allow_bf16 and t.dtype == torch.bfloat16):
        raise _lib.TmfError(f"{name} must be float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _ptr(t):
    return None if t is None else t.data_ptr()


# --------------------------------------------------------------------------------------
# conv3d (+ folded bias) -> BatchNorm3d -> LeakyReLU -> pool      (networks.py:21-53)
# --------------------------------------------------------------------------------------

def pack_weight(weight: torch.Tensor) -> torch.Tensor:
    """(Cout, Cin, k, k, k) reference layout -> tap-major [k^3][Cin][Cout]."""
    return weight.permute(2, 3, 4, 1, 0).contiguous()


def pack_weight_dgrad(weight: torch.Tensor) -> torch.Tensor:
    """Weights of the data-gradient convolution: w'[26-t][co][ci] = w[t][ci][co]."""
    return weight.flip(2, 3, 4).permute(2, 3, 4, 0, 1).contiguous()


def pack_weights_both(weight: torch.Tensor, want_dgrad: bool, want_fwd: bool = True):
    """Forward layout [k^3][Cin][Cout] and (optionally) data-gradient layout [k^3][Cout][Cin] in one launch."""
    cout, cin, k = weight.shape[0], weight.shape[1], weight.shape[2]
    if not (want_fwd or want_dgrad):
        return None, None
    wf = torch.empty((k ** 3, cin, cout), device=weight.device, dtype=_f32) if want_fwd else None
    wd = torch.empty((k ** 3, cout, cin), device=weight.device, dtype=_f32) if want_dgrad else None
    _lib.call("tmf_pack_conv_weights", weight.data_ptr(), _ptr(wf), _ptr(wd), cout, cin, k ** 3, _stream())
    return wf, wd


def unpack_wgrad(dw: torch.Tensor, cout: int, cin: int, k: int) -> torch.Tensor:
    """tap-major [k^3][Cin][Cout] gradient -> reference (Cout, Cin, k, k, k)."""
    return dw.view(k, k, k, cin, cout).permute(4, 3, 0, 1, 2).contiguous()


# Precision of the convolution products is a property of the MODULE that runs them (sNet.set_precision / model.set_precision
# -> sNet.tmf_precision), handed down explicitly to every op and into tmf_snet_desc.precision / .storage_bf16; two models with
# different precisions live side by side in one process.  The two setters below only change the DEFAULT that modules
# without a setting of their own follow (the round-1 API: bench.py --precision, the parity tests).
_PRECISION = "fp32"
_ACT16 = False
_b16 = torch.bfloat16
_CONV_MODES = ("fp32", "fp32x", "bf16")


def set_conv_precision(precision: str) -> None:
    """Process DEFAULT for modules that have no precision of their own: "fp32" (exact-fp32 MFMA everywhere); "bf16": the
    forward, data-gradient and weight-gradient 3x3x3 convolutions round their operands to bf16 and run on the bf16 matrix
    cores with fp32 accumulation (tensors stay fp32); "fp32x": fp32-accurate 3-way bf16 split (six partial products)."""
    global _PRECISION
    if precision not in _CONV_MODES:
        raise ValueError(precision)
    _PRECISION = precision


def get_conv_precision() -> str:
    return _PRECISION


def set_activation_storage(dtype: str) -> None:
    """Process DEFAULT: "fp32" or "bf16" — with conv precision "bf16", keep the sNet activations BETWEEN the conv blocks
    (raw conv outputs z, block outputs, and their gradients) as bf16 tensors — BASELINE configs[2] "bf16 storage".  The
    network input, the BatchNorm statistics / parameters, the 1x1x1 layer and everything after the encoders stay fp32."""
    global _ACT16
    if dtype not in ("fp32", "bf16"):
        raise ValueError(dtype)
    _ACT16 = dtype == "bf16"


def make_precision(conv: str = "fp32", storage: str = "fp32"):
    """(conv mode, bf16 activation storage) as the ops take it; storage "bf16" only has an effect with conv "bf16"."""
    if conv not in _CONV_MODES or storage not in ("fp32", "bf16"):
        raise ValueError((conv, storage))
    return (conv, storage == "bf16" and conv == "bf16")


def resolve_precision(precision=None):
    """None -> the process default; else a make_precision() pair."""
    if precision is None:
        return (_PRECISION, _ACT16 and _PRECISION == "bf16")
    return precision


def activation_storage_bf16(precision=None) -> bool:
    return resolve_precision(precision)[1]


def bf16_conv_capable(cin: int, k: int) -> bool:
    """the 3x3x3 bf16 matrix-core kernels need 8-channel input groups"""
    return k == 3 and cin > 1 and cin % 8 == 0


def pack_weight_bf16(weight: torch.Tensor) -> torch.Tensor:
    """(Cout, Cin, 3,3,3) -> bf16 [27][Cout][Cin] (B operand rows: output channel, K = input channel contiguous)."""
    return weight.permute(2, 3, 4, 0, 1).contiguous().to(torch.bfloat16)


def pack_weights_both_bf16(weight: torch.Tensor, want_dgrad: bool):
    """pack_weight_bf16 and (optionally) pack_weight_dgrad_bf16 of one weight tensor in ONE launch."""
    cout, cin, k = weight.shape[0], weight.shape[1], weight.shape[2]
    wf = torch.empty((k ** 3, cout, cin), device=weight.device, dtype=torch.bfloat16)
    wd = torch.empty((k ** 3, cin, cout), device=weight.device, dtype=torch.bfloat16) if want_dgrad else None
    _lib.call("tmf_pack_conv_weights_bf16", weight.data_ptr(), wf.data_ptr(), _ptr(wd), cout, cin, k ** 3, _stream())
    return wf, wd


def pack_weight_dgrad_bf16(weight: torch.Tensor) -> torch.Tensor:
    """Data-gradient weights, bf16 [27][Cin][Cout]: w'[26-t][ci][co] = w[co][ci][t]."""
    return weight.flip(2, 3, 4).permute(2, 3, 4, 1, 0).contiguous().to(torch.bfloat16)


def split3_bf16(t: torch.Tensor) -> torch.Tensor:
    """Exact decomposition t = hi + mid + lo into three bf16 tensors, stacked on a new leading axis."""
    hi = t.to(torch.bfloat16)
    r1 = t - hi.float()
    mid = r1.to(torch.bfloat16)
    lo = (r1 - mid.float()).to(torch.bfloat16)
    return torch.stack([hi, mid, lo]).contiguous()


def pack_weights_split3(weight: torch.Tensor, want_dgrad: bool):
    """split3_bf16 of the forward layout [27][cout][cin] and (optionally) of the data-gradient layout [27][cin][cout]
    (flipped taps) of one (cout, cin, 3, 3, 3) weight tensor in ONE launch: (3, 27, cout, cin), (3, 27, cin, cout) bf16."""
    cout, cin, k = weight.shape[0], weight.shape[1], weight.shape[2]
    wf = torch.empty((3, k ** 3, cout, cin), device=weight.device, dtype=torch.bfloat16)
    wd = torch.empty((3, k ** 3, cin, cout), device=weight.device, dtype=torch.bfloat16) if want_dgrad else None
    _lib.call("tmf_pack_conv_weights_split3", weight.data_ptr(), wf.data_ptr(), _ptr(wd), cout, cin, k ** 3, _stream())
    return wf, wd


def conv_wino_mode() -> int:
    """tmf_set_option("conv_wino", v) / TMF_CONV_WINO: 0 = direct kernels, 1 = Winograd data gradients, 2 = Winograd forward
    and data gradients (layers with tmf_conv3d_wino_ok), 3 (default) = the weight gradients as well (tmf_conv3d_wgrad_wino_ok)."""
    return _lib.query("tmf_conv_wino_mode")


def wino_ok(cin: int, cout: int) -> bool:
    return bool(_lib.query("tmf_conv3d_wino_ok", cin, cout))


def pack_weights_wino(weight: torch.Tensor, want_fwd: bool = True, want_dgrad: bool = False):
    """Winograd-transformed weights of one (cout, cin, 3, 3, 3) tensor in ONE launch: u_fwd [64][cin/8][2][cout][4] and
    u_dgrad [64][cout/8][2][cin][4] (flipped filter, channel roles swapped); None for a form that is not asked for."""
    cout, cin = weight.shape[0], weight.shape[1]
    # (the library's buffer is the fp32 tensor plus, behind it, its exact 3-way bf16 split for conv3d_winox.hip: the returned
    #  tensors are views of the fp32 part, the storage carries both)
    n = 64 * cin * cout
    nb = _lib.query("tmf_conv3d_wino_weight_bytes", cin, cout) // 4
    uf = torch.empty((nb,), device=weight.device, dtype=_f32)[:n].view(64, cin // 8, 2, cout, 4) if want_fwd else None
    ud = torch.empty((nb,), device=weight.device, dtype=_f32)[:n].view(64, cout // 8, 2, cin, 4) if want_dgrad else None
    _lib.call("tmf_pack_conv_weights_wino", weight.data_ptr(), _ptr(uf), _ptr(ud), cout, cin, _stream())
    return uf, ud


def conv3d_wino_raw(x, u, cin, cout, want_stats):
    """3x3x3 convolution in the Winograd form F(2x2x2, 3x3x3) (csrc/conv3d_wino.hip); u from pack_weights_wino."""
    B, D, H, W = x.shape[:4]
    z = torch.empty((B, D, H, W, cout), device=x.device, dtype=_f32)
    part, nblk = None, 0
    if want_stats:
        nblk = _lib.query("tmf_conv3d_wino_stat_blocks", B, D, H, W)
        part = torch.empty((nblk, 2, cout), device=x.device, dtype=_f32)
    _lib.call("tmf_conv3d_fwd_wino", x.data_ptr(), u.data_ptr(), z.data_ptr(), _ptr(part), B, D, H, W, cin, cout, _stream())
    return z, part, nblk


def wino_p_mode() -> bool:
    """True when the Winograd entries run their persistent one-wave-per-SIMD kernels (the default; TMF_WINO_P=0 or
    tmf_set_option("wino_p", 0): the two-waves-per-SIMD kernels of round 4)."""
    return bool(_lib.query("tmf_wino_p_mode"))


def wgrad_wino_ok(cin: int, cout: int) -> bool:
    return bool(_lib.query("tmf_conv3d_wgrad_wino_ok", cin, cout))


def conv3d_wgrad_wino(x, dz, cin, cout, reference_layout=False):
    """3x3x3 weight gradient in the Winograd form (csrc/conv3d_wino.hip): tap-major [27][cin][cout] or nn.Conv3d's layout."""
    B, D, H, W = x.shape[:4]
    dw = torch.empty((cout, cin, 3, 3, 3) if reference_layout else (27, cin, cout), device=x.device, dtype=_f32)
    nbytes = _lib.query("tmf_conv3d_wgrad_wino_workspace_bytes", B, D, H, W, cin, cout)
    ws = torch.empty((max(nbytes, 16) // 4,), device=x.device, dtype=_f32)
    _lib.call("tmf_conv3d_wgrad_wino", x.data_ptr(), dz.data_ptr(), dw.data_ptr(), ws.data_ptr(), nbytes,
              B, D, H, W, cin, cout, int(reference_layout), _stream())
    return dw


def conv3d_split_raw(x, w3, cin, cout, want_stats):
    """fp32-accurate conv on the bf16 matrix cores; w3 = split3_bf16(packed [27][cout][cin] fp32 weights)."""
    B, D, H, W = x.shape[:4]
    z = torch.empty((B, D, H, W, cout), device=x.device, dtype=_f32)
    part, nblk = None, 0
    if want_stats:
        nblk = _lib.query("tmf_conv3d_split_stat_blocks", B, D, H, W)
        part = torch.empty((nblk, 2, cout), device=x.device, dtype=_f32)
    _lib.call("tmf_conv3d_fwd_split", x.data_ptr(), w3.data_ptr(), z.data_ptr(), _ptr(part),
              B, D, H, W, cin, cout, _stream())
    return z, part, nblk


def conv3d_bf16_raw(x, w_bf16, cin, cout, want_stats, out_bf16=False):
    """z = conv3x3x3(bf16(x), w_bf16) with fp32 accumulation; x may be an fp32 or a bf16 tensor, z is fp32 or
    (out_bf16) bf16; returns (z, stat_partial or None, nblk)."""
    B, D, H, W = x.shape[:4]
    z = torch.empty((B, D, H, W, cout), device=x.device, dtype=_b16 if out_bf16 else _f32)
    part, nblk = None, 0
    if want_stats:
        nblk = _lib.query("tmf_conv3d_bf16_stat_blocks", B, D, H, W)
        part = torch.empty((nblk, 2, cout), device=x.device, dtype=_f32)
    io = (1 if x.dtype == _b16 else 0) | (2 if out_bf16 else 0)
    _lib.call("tmf_conv3d_fwd_bf16_t", x.data_ptr(), w_bf16.data_ptr(), z.data_ptr(), _ptr(part),
              B, D, H, W, cin, cout, io, _stream())
    return z, part, nblk


def conv3d_raw(x, w_packed, cin, cout, ksize, want_stats):
    """z = conv(x, w) on NDHWC x; returns (z, stat_partial or None, nblk)."""
    B, D, H, W = x.shape[:4]
    z = torch.empty((B, D, H, W, cout), device=x.device, dtype=_f32)
    part, nblk = None, 0
    if cin == 1 and ksize == 3:
        if want_stats:
            nblk = _lib.query("tmf_conv3d_c1_stat_blocks", B, D, H, W, cout)
            part = torch.empty((nblk, 2, cout), device=x.device, dtype=_f32)
        _lib.call("tmf_conv3d_c1_fwd", x.data_ptr(), w_packed.data_ptr(), z.data_ptr(), _ptr(part),
                  B, D, H, W, cout, _stream())
    else:
        if want_stats:
            nblk = _lib.query("tmf_conv3d_stat_blocks", B, D, H, W, cin, cout, ksize)
            part = torch.empty((nblk, 2, cout), device=x.device, dtype=_f32)
        _lib.call("tmf_conv3d_fwd", x.data_ptr(), w_packed.data_ptr(), z.data_ptr(), _ptr(part),
                  B, D, H, W, cin, cout, ksize, _stream())
    return z, part, nblk


def conv3d_wgrad_bf16(x, dz, cin, cout, reference_layout=False):
    """weight gradient on the bf16 matrix cores (operands rounded to bf16): tap-major [27][cin][cout], or
    (reference_layout) the nn.Conv3d layout (cout, cin, 3, 3, 3) written directly by the final reduction."""
    B, D, H, W = x.shape[:4]
    dw = torch.empty((cout, cin, 3, 3, 3) if reference_layout else (27, cin, cout), device=x.device, dtype=_f32)
    nbytes = _lib.query("tmf_conv3d_wgrad_bf16_workspace_bytes", B, D, H, W, cin, cout)
    ws = torch.empty((max(nbytes, 16) // 4,), device=x.device, dtype=_f32)
    if x.dtype != dz.dtype:                      # mixed storage (not produced by sNet): widen the bf16 side
        x, dz = x.float(), dz.float()
    _lib.call("tmf_conv3d_wgrad_bf16_t", x.data_ptr(), dz.data_ptr(), dw.data_ptr(), ws.data_ptr(), nbytes,
              B, D, H, W, cin, cout, 1 if x.dtype == _b16 else 0, int(reference_layout), _stream())
    return dw


def conv3d_wgrad(x, dz, cin, cout, ksize, reference_layout=False):
    """weight gradient: tap-major [k^3][cin][cout], or (reference_layout) nn.Conv3d's (cout, cin, k, k, k)."""
    B, D, H, W = x.shape[:4]
    dw = torch.empty((cout, cin, ksize, ksize, ksize) if reference_layout else (ksize ** 3, cin, cout),
                     device=x.device, dtype=_f32)
    if cin == 1 and ksize == 3:
        nbytes = _lib.query("tmf_conv3d_c1_wgrad_workspace_bytes", B, D, H, W, cout)
        ws = torch.empty((max(nbytes, 16) // 4,), device=x.device, dtype=_f32)
        _lib.call("tmf_conv3d_c1_wgrad", x.data_ptr(), dz.data_ptr(), dw.data_ptr(), ws.data_ptr(), nbytes,
                  B, D, H, W, cout, int(reference_layout), _stream())
    else:
        nbytes = _lib.query("tmf_conv3d_wgrad_workspace_bytes", B, D, H, W, cin, cout, ksize)
        ws = torch.empty((max(nbytes, 16) // 4,), device=x.device, dtype=_f32)
        _lib.call("tmf_conv3d_wgrad", x.data_ptr(), dz.data_ptr(), dw.data_ptr(), ws.data_ptr(), nbytes,
                  B, D, H, W, cin, cout, ksize, int(reference_layout), _stream())
    return dw


class ConvBnActPool(torch.autograd.Function):
    """One sNet block: Conv3d(k, same padding) -> BatchNorm3d -> LeakyReLU -> {none,max,avg} 2x2x2 pool.

    The conv bias is never added to the activations: ahead of BatchNorm it only shifts the
    batch mean, so it is folded into running_mean (train) or into the BN shift (eval).
    Its train-mode gradient is exactly zero in exact arithmetic (the reference returns
    rounding noise there); we return zeros.
    """

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, running_mean, running_var,
                training, momentum, eps, slope, pool, out_bf16=False, precision=None):
        mode, act16 = resolve_precision(precision)
        x = _chk(x, "x", allow_bf16=True)
        cout, cin, k = weight.shape[0], weight.shape[1], weight.shape[2]
        B, D, H, W, C = x.shape
        if C != cin:
            raise _lib.TmfError(f"conv expects {cin} input channels, got {C}")
        weight = _chk(weight, "weight")
        bf16 = mode if (mode != "fp32" and k == 3 and cin % 8 == 0 and cin > 1) else False
        z16 = bf16 == "bf16" and act16           # raw conv output (and dz) stored as bf16
        if x.dtype == _b16 and bf16 != "bf16":
            x = x.float()                         # only the bf16 kernels read bf16 tensors
        if out_bf16 and not z16:
            raise _lib.TmfError("a bf16 block output needs conv precision 'bf16' with bf16 activation storage")
        wf = wd = None
        # Winograd form of the fp32 products (tmf_set_option("conv_wino", ..)): the same choice per layer and
        # direction as the whole-encoder path (snet_path.hip make_plan), so the two stay bit-identical
        wino = conv_wino_mode() if (not bf16 and k == 3 and cin > 1) else 0
        wino_f = wino >= 2 and wino_ok(cin, cout)
        wino_d = wino >= 1 and wino_ok(cout, cin)
        ctx.wino_w = wino >= 3 and wgrad_wino_ok(cin, cout)
        if not bf16:                      # both weight layouts in one launch; the dgrad one is kept for backward
            want_d = ctx.needs_input_grad[0]
            wf, wd = pack_weights_both(weight, want_d and not wino_d, want_fwd=not wino_f)
            if wino_f or (wino_d and want_d):
                uf, ud = pack_weights_wino(weight, wino_f, wino_d and want_d)
                wf = uf if wino_f else wf
                wd = ud if (wino_d and want_d) else wd
        elif bf16 == "bf16":
            wf, wd = pack_weights_both_bf16(weight, ctx.needs_input_grad[0] and cout % 8 == 0)
        elif ctx.needs_input_grad[0] and cout % 8 == 0:      # fp32x: both split layouts in one launch
            wf, wd = pack_weights_split3(weight, True)
        else:
            wf, wd = pack_weights_split3(weight, False)

        def conv(stats):
            if bf16 == "bf16":
                return conv3d_bf16_raw(x, wf, cin, cout, stats, out_bf16=z16)
            if bf16 == "fp32x":
                return conv3d_split_raw(x, wf, cin, cout, stats)
            if wino_f:
                return conv3d_wino_raw(x, wf, cin, cout, stats)
            return conv3d_raw(x, wf, cin, cout, k, stats)

        dev = x.device
        mean = torch.empty(cout, device=dev, dtype=_f32)
        invstd = torch.empty(cout, device=dev, dtype=_f32)
        scale = torch.empty(cout, device=dev, dtype=_f32)
        shift = torch.empty(cout, device=dev, dtype=_f32)
        s = _stream()
        pc = _lib.pool_code(pool)
        if training:
            z, part, nblk = conv(True)
            _lib.call("tmf_bn_finalize", part.data_ptr(), nblk, cout, float(B * D * H * W),
                      gamma.data_ptr(), beta.data_ptr(), _ptr(bias), _ptr(running_mean), _ptr(running_var),
                      float(momentum), float(eps), mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(),
                      shift.data_ptr(), s)
        else:
            z, _, _ = conv(False)
            _lib.call("tmf_bn_eval_coeffs", gamma.data_ptr(), beta.data_ptr(), _ptr(bias), running_mean.data_ptr(),
                      running_var.data_ptr(), float(eps), cout, scale.data_ptr(), shift.data_ptr(), s)
            # xhat = (z + bias - running_mean) * invstd, written as (z - mean) * invstd
            invstd = torch.rsqrt(running_var + eps)
            mean = running_mean - bias if bias is not None else running_mean.clone()
        odt = _b16 if out_bf16 else _f32
        if pc == _lib.POOL_NONE:
            out = torch.empty((B, D, H, W, cout), device=dev, dtype=odt)
        else:
            out = torch.empty((B, D // 2, H // 2, W // 2, cout), device=dev, dtype=odt)
        io = (1 if z16 else 0) | (2 if out_bf16 else 0)
        if out.numel() > 0:
            _lib.call("tmf_bn_act_pool_fwd_t", z.data_ptr(), scale.data_ptr(), shift.data_ptr(), out.data_ptr(),
                      B, D, H, W, cout, pc, float(slope), io, s)
        ctx.save_for_backward(x, weight if wd is None else wd, z, scale, shift, mean, invstd)
        ctx.cfg = (training, float(slope), pc, cin, cout, k, bias is not None)
        ctx.bf16 = bf16
        ctx.packed_dgrad = wd is not None
        ctx.wino_d = bool(wino_d and wd is not None)
        ctx.io = io
        return out

    @staticmethod
    def backward(ctx, dout):
        x, weight, z, scale, shift, mean, invstd = ctx.saved_tensors
        training, slope, pc, cin, cout, k, has_bias = ctx.cfg
        B, D, H, W, _ = x.shape
        dev = x.device
        s = _stream()
        io = ctx.io
        dout = _chk(dout, "grad_output", allow_bf16=True)
        if (dout.dtype == _b16) != bool(io & 2):
            dout = dout.to(_b16 if io & 2 else _f32)
        nblk = _lib.query("tmf_bn_act_pool_bwd_blocks", B, D, H, W, cout, pc)
        part = torch.empty((nblk, 2, cout), device=dev, dtype=_f32)
        _lib.call("tmf_bn_act_pool_bwd_reduce_t", z.data_ptr(), dout.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                  mean.data_ptr(), invstd.data_ptr(), part.data_ptr(), B, D, H, W, cout, pc, slope, io, s)
        dgamma = torch.empty(cout, device=dev, dtype=_f32)
        dbeta = torch.empty(cout, device=dev, dtype=_f32)
        coef = torch.empty((2, cout), device=dev, dtype=_f32)
        _lib.call("tmf_bn_bwd_finalize", part.data_ptr(), nblk, cout, float(B * D * H * W),
                  dgamma.data_ptr(), dbeta.data_ptr(), coef.data_ptr(), s)
        if training:
            dbias = torch.zeros(cout, device=dev, dtype=_f32) if has_bias else None
        else:
            coef.zero_()                      # eval-mode BN is affine: dz = scale * dy
            dbias = scale * dbeta if has_bias else None
        dz = torch.empty_like(z)
        _lib.call("tmf_bn_act_pool_bwd_apply_t", z.data_ptr(), dout.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                  mean.data_ptr(), invstd.data_ptr(), coef.data_ptr(), dz.data_ptr(), B, D, H, W, cout, pc, slope, io, s)
        dweight = None
        if ctx.needs_input_grad[1]:
            if ctx.bf16 == "bf16":
                dweight = conv3d_wgrad_bf16(x, dz, cin, cout, reference_layout=True)
            elif ctx.wino_w:
                dweight = conv3d_wgrad_wino(x, dz, cin, cout, reference_layout=True)
            else:
                dweight = conv3d_wgrad(x, dz, cin, cout, k, reference_layout=True)
        dx = None
        if ctx.needs_input_grad[0]:
            if ctx.bf16 == "bf16" and cout % 8 == 0:
                dx, _, _ = conv3d_bf16_raw(dz, weight, cout, cin, False, out_bf16=x.dtype == _b16)    # weight = packed wd
            elif ctx.bf16 == "fp32x" and cout % 8 == 0:
                dx, _, _ = conv3d_split_raw(dz, weight, cout, cin, False)             # weight = packed split wd
            elif ctx.wino_d:
                dx, _, _ = conv3d_wino_raw(dz, weight, cout, cin, False)              # weight = packed Winograd u_dgrad
            else:
                dzf = dz if dz.dtype == _f32 else dz.float()
                dx, _, _ = conv3d_raw(dzf, weight if ctx.packed_dgrad else pack_weight_dgrad(weight), cout, cin, k, False)
            if dx.dtype != x.dtype:
                dx = dx.to(x.dtype)
        return (dx, dweight, dbias, dgamma, dbeta, None, None, None, None, None, None, None, None, None)


class Conv1BnPool(torch.autograd.Function):
    """First sNet block (Cin = 1, 3x3x3, max pool) with the conv output never written to HBM: the statistics,
    forward, backward-reduce and weight-gradient passes each recompute it from the input volume."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, running_mean, running_var, training, momentum, eps, slope,
                out_bf16=False, precision=None):
        mode, _act16 = resolve_precision(precision)
        x = _chk(x, "x")
        B, D, H, W, _ = x.shape
        C = weight.shape[0]
        wp = pack_weight(_chk(weight, "weight")).view(27, C)
        dev = x.device
        mean = torch.empty(C, device=dev, dtype=_f32)
        invstd = torch.empty(C, device=dev, dtype=_f32)
        scale = torch.empty(C, device=dev, dtype=_f32)
        shift = torch.empty(C, device=dev, dtype=_f32)
        s = _stream()
        sfx = "_bf16" if mode == "bf16" else ""            # both products on the bf16 matrix cores (opt-in mode)
        gram = None
        if training:
            nblk = _lib.query("tmf_c1_blocks", B, D, H, W, C)
            part = torch.empty((nblk, 2, C), device=dev, dtype=_f32)
            gbytes = _lib.query("tmf_c1_gram_bytes" + sfx, B, D, H, W, C) if mode in ("fp32", "bf16") else 0   # (bf16: "c1_gram" 2 only)
            if gbytes:                 # pair sums + the tap Gram matrix of the volume: backward then needs one pass
                gram = torch.empty(gbytes // 8, device=dev, dtype=torch.float64)   # (bf16: of the volume rounded to bf16)
                _lib.call("tmf_c1_stats_g" + sfx, x.data_ptr(), wp.data_ptr(), part.data_ptr(), gram.data_ptr(), gbytes, B, D, H, W, C, s)
                rows = 2
            else:                      # the recomputing pass (bf16: its own; fp32x keeps it: DESIGN 3.16)
                _lib.call("tmf_c1_stats" + sfx, x.data_ptr(), wp.data_ptr(), part.data_ptr(), B, D, H, W, C, s)
                rows = nblk
            _lib.call("tmf_bn_finalize", part.data_ptr(), rows, C, float(B * D * H * W),
                      gamma.data_ptr(), beta.data_ptr(), _ptr(bias), _ptr(running_mean), _ptr(running_var),
                      float(momentum), float(eps), mean.data_ptr(), invstd.data_ptr(), scale.data_ptr(),
                      shift.data_ptr(), s)
        else:
            _lib.call("tmf_bn_eval_coeffs", gamma.data_ptr(), beta.data_ptr(), _ptr(bias), running_mean.data_ptr(),
                      running_var.data_ptr(), float(eps), C, scale.data_ptr(), shift.data_ptr(), s)
            invstd = torch.rsqrt(running_var + eps)
            mean = running_mean - bias if bias is not None else running_mean.clone()
        if out_bf16 and not sfx:
            raise _lib.TmfError("a bf16 block output needs conv precision 'bf16'")
        out = torch.empty((B, D // 2, H // 2, W // 2, C), device=dev, dtype=_b16 if out_bf16 else _f32)
        p16 = (int(out_bf16),) if sfx else ()
        if out.numel() > 0:
            _lib.call("tmf_c1_bn_pool_fwd" + sfx, x.data_ptr(), wp.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                      out.data_ptr(), B, D, H, W, C, float(slope), *p16, s)
        ctx.save_for_backward(x, wp, scale, shift, mean, invstd)
        ctx.cfg = (training, float(slope), C, bias is not None, sfx, p16)
        ctx.gram = gram
        return out

    @staticmethod
    def backward(ctx, dout):
        x, wp, scale, shift, mean, invstd = ctx.saved_tensors
        training, slope, C, has_bias, sfx, p16 = ctx.cfg
        B, D, H, W, _ = x.shape
        dev = x.device
        s = _stream()
        dout = _chk(dout, "grad_output", allow_bf16=True)
        want16 = bool(p16 and p16[0])
        if (dout.dtype == _b16) != want16:
            dout = dout.to(_b16 if want16 else _f32)
        if ctx.gram is not None and training and ctx.needs_input_grad[1] and not ctx.needs_input_grad[0]:
            # one pass over the volume: BatchNorm sums and D = x (*) dy together, dw from the forward's Gram data
            nbytes = _lib.query("tmf_c1_bwd_fused_workspace_bytes", B, D, H, W, C)
            ws = torch.empty((max(nbytes, 16) // 4,), device=dev, dtype=_f32)
            dweight = torch.empty((C, 1, 3, 3, 3), device=dev, dtype=_f32)
            dgamma = torch.empty(C, device=dev, dtype=_f32)
            dbeta = torch.empty(C, device=dev, dtype=_f32)
            _lib.call("tmf_c1_bwd_fused" + sfx, x.data_ptr(), wp.data_ptr(), scale.data_ptr(), shift.data_ptr(), mean.data_ptr(),
                      invstd.data_ptr(), dout.data_ptr(), ctx.gram.data_ptr(), dweight.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                      ws.data_ptr(), nbytes, B, D, H, W, C, slope, *p16, _lib.DW_REFERENCE, s)
            dbias = torch.zeros(C, device=dev, dtype=_f32) if has_bias else None
            return (None, dweight, dbias, dgamma, dbeta, None, None, None, None, None, None, None, None)
        nblk = _lib.query("tmf_c1_blocks", B, D, H, W, C)
        part = torch.empty((nblk, 2, C), device=dev, dtype=_f32)
        _lib.call("tmf_c1_bwd_reduce" + sfx, x.data_ptr(), wp.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                  mean.data_ptr(), invstd.data_ptr(), dout.data_ptr(), part.data_ptr(), B, D, H, W, C, slope, *p16, s)
        dgamma = torch.empty(C, device=dev, dtype=_f32)
        dbeta = torch.empty(C, device=dev, dtype=_f32)
        coef = torch.empty((2, C), device=dev, dtype=_f32)
        _lib.call("tmf_bn_bwd_finalize", part.data_ptr(), nblk, C, float(B * D * H * W),
                  dgamma.data_ptr(), dbeta.data_ptr(), coef.data_ptr(), s)
        if training:
            dbias = torch.zeros(C, device=dev, dtype=_f32) if has_bias else None
        else:
            coef.zero_()
            dbias = scale * dbeta if has_bias else None
        dweight = None
        if ctx.needs_input_grad[1]:
            nbytes = _lib.query("tmf_c1_bwd_wgrad_workspace_bytes", B, D, H, W, C)
            ws = torch.empty((max(nbytes, 16) // 4,), device=dev, dtype=_f32)
            dweight = torch.empty((C, 1, 3, 3, 3), device=dev, dtype=_f32)       # written in the reference layout
            _lib.call("tmf_c1_bwd_wgrad" + sfx, x.data_ptr(), wp.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                      mean.data_ptr(), invstd.data_ptr(), coef.data_ptr(), dout.data_ptr(), dweight.data_ptr(),
                      ws.data_ptr(), nbytes, B, D, H, W, C, slope, *p16, _lib.DW_REFERENCE, s)
        if ctx.needs_input_grad[0]:
            raise _lib.TmfError("the fused first block has no data gradient (the network input needs none)")
        return (None, dweight, dbias, dgamma, dbeta, None, None, None, None, None, None, None, None)


def conv_bn_act_pool_eval_fused(x, weight, bias, gamma, beta, running_mean, running_var, eps, slope, pool):
    """Inference form of a block (no autograd graph): conv + folded BatchNorm + LeakyReLU + pool in ONE kernel."""
    x, weight = _chk(x, "x"), _chk(weight, "weight")
    cout, cin, k = weight.shape[0], weight.shape[1], weight.shape[2]
    B, D, H, W, C = x.shape
    if C != cin:
        raise _lib.TmfError(f"conv expects {cin} input channels, got {C}")
    dev = x.device
    scale = torch.empty(cout, device=dev, dtype=_f32)
    shift = torch.empty(cout, device=dev, dtype=_f32)
    s = _stream()
    _lib.call("tmf_bn_eval_coeffs", gamma.data_ptr(), beta.data_ptr(), _ptr(bias), running_mean.data_ptr(),
              running_var.data_ptr(), float(eps), cout, scale.data_ptr(), shift.data_ptr(), s)
    pc = _lib.pool_code(pool)
    shape = (B, D, H, W, cout) if pc == _lib.POOL_NONE else (B, D // 2, H // 2, W // 2, cout)
    out = torch.empty(shape, device=dev, dtype=_f32)
    if out.numel() > 0:
        if k == 3 and conv_wino_mode() >= 2 and wino_ok(cin, cout) and pool != "avg":       # as tmf_snet_eval_fwd chooses
            uf, _ = pack_weights_wino(weight, True, False)
            _lib.call("tmf_conv3d_fwd_wino_affine", x.data_ptr(), uf.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                      out.data_ptr(), B, D, H, W, cin, cout, pc, float(slope), s)
        else:
            _lib.call("tmf_conv3d_fwd_affine", x.data_ptr(), pack_weight(weight).data_ptr(), scale.data_ptr(),
                      shift.data_ptr(), out.data_ptr(), B, D, H, W, cin, cout, k, pc, float(slope), s)
    return out


FUSE_EVAL_BLOCKS = os.environ.get("TMF_FUSE_EVAL", "1") != "0"


def conv_bn_act_pool(x, weight, bias, gamma, beta, running_mean, running_var, training,
                     momentum=0.1, eps=1e-5, slope=0.01, pool=None, out_bf16=False, precision=None):
    """precision: make_precision(conv, storage) of the calling module, None = the process default."""
    precision = resolve_precision(precision)
    mode, act16 = precision
    needs_graph = torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad or gamma.requires_grad)
    if (FUSE_EVAL_BLOCKS and not training and not needs_graph and mode == "fp32" and weight.shape[1] > 1
            and weight.shape[1] % 4 == 0 and weight.shape[0] % 4 == 0):
        return conv_bn_act_pool_eval_fused(x, weight, bias, gamma, beta, running_mean, running_var, eps, slope, pool)
    if (weight.shape[1] == 1 and weight.shape[2] == 3 and pool == "max" and not x.requires_grad):
        if out_bf16 and mode != "bf16":
            raise _lib.TmfError("a bf16 block output needs conv precision 'bf16'")
        return Conv1BnPool.apply(x, weight, bias, gamma, beta, running_mean, running_var,
                                 training, momentum, eps, slope, out_bf16 and mode == "bf16", precision)
    can16 = out_bf16 and act16 and bf16_conv_capable(weight.shape[1], weight.shape[2])
    y = ConvBnActPool.apply(x, weight, bias, gamma, beta, running_mean, running_var,
                            training, momentum, eps, slope, pool, can16, precision)
    return y.to(_b16) if (out_bf16 and not can16) else y


# --------------------------------------------------------------------------------------
# whole encoder in one call per pass                                   (networks.py:55-61)
# --------------------------------------------------------------------------------------

# False / TMF_SNET_C=0: every block is its own autograd.Function issued from Python (kept for A/B runs and tests)
SNET_ONE_CALL = os.environ.get("TMF_SNET_C", "1") != "0"


def snet_one_call_supported(B, D, H, W, dim, precision=None):
    return (SNET_ONE_CALL and resolve_precision(precision)[0] in ("fp32", "bf16", "fp32x") and dim >= 32 and dim % 32 == 0
            and min(D, H, W) >= 16 and B > 0)


def snet_eval_one_call(vol, dim, eps, slope, blocks, precision=None, algo=None):
    """Eval-mode sNet forward as ONE library call (tmf_snet_eval_fwd): no autograd graph (val_step runs under no_grad).
    blocks: 7 x (conv weight, conv bias | None, bn weight, bn bias, running_mean, running_var); precision: fp32 (a block is
    one kernel) or bf16 [+ bf16 storage] (the block-by-block launches, enqueued by one call)."""
    import ctypes as C
    vol = _chk(vol, "vol")
    B, _, D, H, W = vol.shape
    mode, act16 = resolve_precision(precision)
    desc = _lib.SnetDesc(B=B, D=D, H=H, W=W, dim=dim, precision={"fp32": 0, "bf16": 1}[mode], storage_bf16=int(act16),
                         flags=snet_algo_flags(algo))
    prm = _lib.SnetParams()
    for l, (w, b, g, be, rm, rv) in enumerate(blocks):
        desc.eps[l], desc.slope[l], desc.momentum[l] = eps[l], slope[l], 0.0
        prm.weight[l], prm.bias[l], prm.gamma[l], prm.beta[l] = w.data_ptr(), _ptr(b), g.data_ptr(), be.data_ptr()
        prm.running_mean[l], prm.running_var[l] = rm.data_ptr(), rv.data_ptr()
    nws = _lib.query("tmf_snet_eval_workspace_bytes", C.byref(desc))
    ws = torch.empty(nws, device=vol.device, dtype=torch.uint8)
    out = torch.empty((B, D // 16, H // 16, W // 16, dim), device=vol.device, dtype=_f32)
    _lib.call("tmf_snet_eval_fwd", C.byref(desc), vol.data_ptr(), C.byref(prm), ws.data_ptr(), nws, out.data_ptr(), _stream())
    return out


# Every whole-pass autograd node below (SNetTrain, FusionTrain, HeadsAD, HeadsCNN) writes ALL of its parameter gradients
# into ONE flat buffer and hands autograd views of it.  A data-parallel wrapper (parallel.GradAllReduce) registers itself
# here as a consumer and is told, at the end of each node's backward, about that buffer: it then all-reduces the buffer IN
# PLACE on its own stream (no per-parameter hooks, no pack copies; `param.grad` ends up as a view of the reduced buffer).
# A segment (start, stop, event, when) says that elements [start, stop) are final behind `event` (None: behind everything
# the CURRENT stream has been handed so far) and when their collective should be handed to RCCL: "now" (an encoder's deep
# blocks: the event is recorded in the middle of its backward, conv2 / conv1 still run behind it), "next" (heads, fusion:
# they report in during the launch-bound start of backward, where the host time of an enqueue is idle GPU — they go out
# together with the next "now" range) or "end" (an encoder's shallow blocks: the end of backward).  The set holds consumers WEAKLY: with no
# wrapper alive nothing is published and nothing is kept (k-fold training builds and drops one model per fold).
import weakref

_FLAT_GRAD_CONSUMERS = weakref.WeakSet()


def add_flat_grad_consumer(consumer) -> None:
    """consumer.tmf_flat_grads(flat, param_ptrs, views, segments) is called at the end of every whole-pass backward."""
    _FLAT_GRAD_CONSUMERS.add(consumer)


def remove_flat_grad_consumer(consumer) -> None:
    _FLAT_GRAD_CONSUMERS.discard(consumer)


def _publish_flat_grads(flat, param_ptrs, views, segments) -> None:
    """param_ptrs[i]: data pointer of the parameter whose gradient is views[i] (None: that input got no gradient).
    A consumer must not KEEP a reference to a view: autograd adopts an incoming gradient as ``.grad`` without a copy only
    while it holds the last reference."""
    for c in list(_FLAT_GRAD_CONSUMERS):
        c.tmf_flat_grads(flat, param_ptrs, views, segments)


def snet_algo_flags(algo=None) -> int:
    """The algorithm word of a tmf_snet_desc (include/tmf_hip.h: TMF_SNET_ALGO | ...).  algo None: the process options of the
    moment (tmf_set_option / TMF_* environment), pinned for the call — a backward then runs the plan its forward laid out whatever
    happens to the options in between; a dict {conv_wino: 0..3, wino_p: 0|1, wino_x: 0|1, c1_gram: 0|1|2, c1_split: 0|1} (missing keys: the process
    option) is ONE module's own choice (sNet.set_algorithm): two models with different settings live side by side."""
    f = _lib.query("tmf_snet_algo_flags")
    if algo:
        bad = set(algo) - {"conv_wino", "wino_p", "wino_x", "c1_gram", "c1_split"}
        if bad:
            raise ValueError(f"unknown algorithm option(s) {sorted(bad)}")
        if "conv_wino" in algo:
            if algo["conv_wino"] not in (0, 1, 2, 3):
                raise ValueError("conv_wino must be 0, 1, 2 or 3")
            f = (f & ~(3 << 9)) | (int(algo["conv_wino"]) << 9)
        for key, bit in (("wino_p", 0x800), ("wino_x", 0x1000), ("c1_gram", 0x2000), ("c1_split", 0x8000)):
            if key in algo:
                f = (f | bit) if algo[key] else (f & ~bit)
        if "c1_gram" in algo:                            # 2: the bf16 mode's first block through the Gram matrix as well
            f = (f | 0x4000) if algo["c1_gram"] == 2 else (f & ~0x4000)
    return f


class SNetTrain(torch.autograd.Function):
    """Train-mode sNet forward / backward as ONE library call each (tmf_snet_train_fwd / _bwd: csrc/snet_path.hip): the
    same kernels in the same order as the block-by-block path, every intermediate tensor inside one workspace tensor,
    weight gradients written in the reference layout.  forward(vol (B,1,D,H,W), cfg, buffers, *params) with params =
    7 x (conv weight, conv bias | None, bn weight, bn bias) -> (B, d, h, w, dim) channels-last."""

    @staticmethod
    def forward(ctx, vol, cfg, buffers, *params):
        import ctypes as C
        vol = _chk(vol, "vol")
        dim, momentum, eps, slope = cfg[:4]
        mode, act16 = resolve_precision(cfg[4] if len(cfg) > 4 else None)
        B, _, D, H, W = vol.shape
        desc = _lib.SnetDesc(B=B, D=D, H=H, W=W, dim=dim, precision={"fp32": 0, "bf16": 1, "fp32x": 2}[mode], storage_bf16=int(act16),
                             flags=(_lib.SNET_ALONE if (len(cfg) > 5 and cfg[5]) else 0) | snet_algo_flags(cfg[6] if len(cfg) > 6 else None))
        prm = _lib.SnetParams()
        for l in range(7):
            desc.momentum[l], desc.eps[l], desc.slope[l] = momentum[l], eps[l], slope[l]
            w, b, g, be = params[4 * l:4 * l + 4]
            rm, rv = buffers[l]
            for t, name in ((w, "conv weight"), (g, "BatchNorm weight"), (be, "BatchNorm bias")):
                if not (t.is_cuda and t.dtype == _f32 and t.is_contiguous()):
                    raise _lib.TmfError(f"sNet block {l}: {name} must be a contiguous float32 tensor on the HIP device")
            prm.weight[l], prm.bias[l], prm.gamma[l], prm.beta[l] = w.data_ptr(), _ptr(b), g.data_ptr(), be.data_ptr()
            prm.running_mean[l], prm.running_var[l] = _ptr(rm), _ptr(rv)
        nsaved = _lib.query("tmf_snet_saved_bytes", C.byref(desc))
        if nsaved == 0:
            raise _lib.TmfError("tmf_snet_saved_bytes: " + (_lib.load().tmf_last_error_string() or b"").decode())
        saved = torch.empty(nsaved, device=vol.device, dtype=torch.uint8)
        out = torch.empty((B, D // 16, H // 16, W // 16, dim), device=vol.device, dtype=_f32)
        _lib.call("tmf_snet_train_fwd", C.byref(desc), vol.data_ptr(), C.byref(prm), saved.data_ptr(), nsaved,
                  out.data_ptr(), _stream())
        ctx.save_for_backward(vol, saved)
        ctx.desc = desc
        ctx.shapes = [None if p is None else p.shape for p in params]
        ctx.param_ptrs = [None if p is None else p.data_ptr() for p in params]
        # the gradient buffer, its views and the gradient table of backward, set up while the GPU is busy (see FusionTrain)
        ctx.bwd = SNetTrain._prepare_backward(ctx.shapes, ctx.needs_input_grad, vol.device) if any(ctx.needs_input_grad[3:]) else None
        return out

    @staticmethod
    def _prepare_backward(shapes, need, dev):
        sizes = [0 if s is None else s.numel() for s in shapes]
        flat = torch.empty(sum(sizes), device=dev, dtype=_f32)        # all 28 gradients in one allocation
        # layout [shallow blocks 0 .. DEEP_FROM-1 | deep blocks DEEP_FROM .. 6 | the seven conv-bias gradients]: the bias
        # gradients (exact zeros) sit back to back at the end — the library fills them with ONE memset at the START of
        # backward — so [deep | biases] is one contiguous range that is final at `deep_event`
        shallow = [i for i in range(4 * _lib.SNET_DEEP_FROM) if i % 4 != 1]
        order = shallow + [i for i in range(4 * _lib.SNET_DEEP_FROM, len(sizes)) if i % 4 != 1] + \
            [i for i in range(len(sizes)) if i % 4 == 1]
        parts = dict(zip(order, flat.split([sizes[i] for i in order])))
        o_deep = sum(sizes[i] for i in shallow)
        grads = [None if s is None else (parts[i] if len(s) == 1 else parts[i].view(s)) for i, s in enumerate(shapes)]
        ptr = [parts[i].data_ptr() for i in range(len(sizes))]
        g = _lib.SnetGrads()
        for l in range(7):
            g.dweight[l] = ptr[4 * l] if need[3 + 4 * l] else None
            g.dbias[l] = None if shapes[4 * l + 1] is None else ptr[4 * l + 1]
            g.dgamma[l], g.dbeta[l] = ptr[4 * l + 2], ptr[4 * l + 3]
        # an event behind the last kernel of the deep blocks (conv3.0 .. conv4.3): a data-parallel wrapper starts the
        # all-reduce of their gradients there, under the backward of conv2 / conv1, instead of after this call
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))               # (creates the handle; the library re-records it in backward)
        g.deep_event = ev.cuda_event
        return flat, grads, g, ev, o_deep

    @staticmethod
    def backward(ctx, dout):
        import ctypes as C
        vol, saved = ctx.saved_tensors
        desc = ctx.desc
        if ctx.needs_input_grad[0]:
            raise _lib.TmfError("the one-call sNet has no data gradient (the network input needs none)")
        dout = _chk(dout, "grad_output")
        prep = ctx.bwd if ctx.bwd is not None else SNetTrain._prepare_backward(ctx.shapes, ctx.needs_input_grad, vol.device)
        ctx.bwd = None
        flat, grads, g, ev, o_deep = prep
        nscr = _lib.query("tmf_snet_bwd_scratch_bytes", C.byref(desc))
        scratch = torch.empty(nscr, device=vol.device, dtype=torch.uint8)
        _lib.call("tmf_snet_train_bwd", C.byref(desc), vol.data_ptr(), saved.data_ptr(), saved.numel(), dout.data_ptr(),
                  C.byref(g), scratch.data_ptr(), nscr, _stream())
        out = [None, None, None]
        for i, gr in enumerate(grads):
            out.append(gr if (gr is not None and ctx.needs_input_grad[3 + i]) else None)
        if _FLAT_GRAD_CONSUMERS and all(ctx.needs_input_grad[3 + i] for i, gr in enumerate(grads) if gr is not None):
            # shallow blocks: final when this call's last kernel is; deep blocks + bias zeros: final at the event
            _publish_flat_grads(flat, ctx.param_ptrs, grads, [(o_deep, flat.numel(), ev, "now"), (0, o_deep, None, "end")])
        return tuple(out)


# --------------------------------------------------------------------------------------
# fused cross attention                                               (networks.py:166-174)
# --------------------------------------------------------------------------------------

class CrossAttention(torch.autograd.Function):
    """out[b, n, (h d)] = softmax(q k^T * scale) v with q: (B, N, h*d), kv: (B, M, 2*h*d)
    (the raw to_q / to_kv outputs: no rearrange, no chunk copy)."""

    @staticmethod
    def forward(ctx, q, kv, heads, scale):
        q, kv = _chk(q, "q"), _chk(kv, "kv")
        B, N, inner = q.shape
        M = kv.shape[1]
        if kv.shape[2] != 2 * inner or inner % heads:
            raise _lib.TmfError(f"attention shapes: q {tuple(q.shape)} kv {tuple(kv.shape)} heads {heads}")
        dh = inner // heads
        out = torch.empty((B, N, inner), device=q.device, dtype=_f32)
        lse = torch.empty((B, heads, N), device=q.device, dtype=_f32)
        _lib.call("tmf_xattn_fwd", q.data_ptr(), kv.data_ptr(), kv.data_ptr() + inner * 4, out.data_ptr(),
                  lse.data_ptr(), B, heads, N, M, dh, inner, 2 * inner, float(scale), _stream())
        ctx.save_for_backward(q, kv, out, lse)
        ctx.cfg = (heads, float(scale))
        return out

    @staticmethod
    def backward(ctx, dout):
        q, kv, out, lse = ctx.saved_tensors
        heads, scale = ctx.cfg
        B, N, inner = q.shape
        M = kv.shape[1]
        dh = inner // heads
        dout = _chk(dout, "grad_output")
        dq = torch.empty_like(q)
        dkv = torch.empty_like(kv)
        _lib.call("tmf_xattn_bwd", q.data_ptr(), kv.data_ptr(), kv.data_ptr() + inner * 4, out.data_ptr(),
                  lse.data_ptr(), dout.data_ptr(), dq.data_ptr(), dkv.data_ptr(), dkv.data_ptr() + inner * 4,
                  B, heads, N, M, dh, inner, 2 * inner, 2 * inner, scale, _stream())
        return dq, dkv, None, None


def cross_attention(q, kv, heads, scale):
    return CrossAttention.apply(q, kv, heads, scale)


# --------------------------------------------------------------------------------------
# LayerNorm                                                          (networks.py:117,219)
# --------------------------------------------------------------------------------------

class LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, residual):
        x = _chk(x, "x")
        dim = x.shape[-1]
        rows = x.numel() // dim
        y = torch.empty_like(x)
        mean = torch.empty(rows, device=x.device, dtype=_f32)
        rstd = torch.empty(rows, device=x.device, dtype=_f32)
        res = _chk(residual, "residual") if residual is not None else None
        _lib.call("tmf_layernorm_fwd", x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _ptr(res), y.data_ptr(),
                  mean.data_ptr(), rstd.data_ptr(), rows, dim, float(eps), _stream())
        ctx.save_for_backward(x, gamma, mean, rstd)
        ctx.has_res = residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, rstd = ctx.saved_tensors
        dim = x.shape[-1]
        rows = x.numel() // dim
        dy = _chk(dy, "grad_output")
        dx = torch.empty_like(x)
        nblk = _lib.query("tmf_layernorm_bwd_blocks", rows, dim)
        part = torch.empty((nblk, 2, dim), device=x.device, dtype=_f32)
        s = _stream()
        _lib.call("tmf_layernorm_bwd", x.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                  dy.data_ptr(), dx.data_ptr(), part.data_ptr(), rows, dim, s)
        gb = torch.empty((2, dim), device=x.device, dtype=_f32)
        _lib.call("tmf_colsum_finalize", part.data_ptr(), nblk, 2 * dim, gb.data_ptr(), s)
        return dx, gb[0], gb[1], None, (dy if ctx.has_res else None)


def layer_norm(x, gamma, beta, eps=1e-5, residual=None):
    """LayerNorm(x) * gamma + beta (+ residual, fused into the same pass)."""
    return LayerNorm.apply(x, gamma, beta, eps, residual)


# --------------------------------------------------------------------------------------
# fused transformer block: every nn.Linear is one launch that also does the LayerNorm / bias / GELU /
# residual around it (csrc/token_gemm.hip).                         (networks.py:114-175, 215-230)
# --------------------------------------------------------------------------------------

def tok_linear_fwd(x, w, bias=None, residual=None, ln=None, gelu=False, keep_ln_out=False):
    """y = [GELU](LayerNorm?(x) @ w.T + bias) + residual on 2-D row-major fp32 tensors.
    ln = (gamma, beta, eps).  Returns (y, ln_saved, pre): ln_saved = (mean, rstd, normalised rows | None)."""
    R, K = x.shape
    nout = w.shape[0]
    y = torch.empty((R, nout), device=x.device, dtype=_f32)
    mean = rstd = ln_out = pre = None
    g = b = None
    eps = 0.0
    if ln is not None:
        g, b, eps = ln
        mean = torch.empty(R, device=x.device, dtype=_f32)
        rstd = torch.empty(R, device=x.device, dtype=_f32)
        if keep_ln_out:
            ln_out = torch.empty((R, K), device=x.device, dtype=_f32)
    if gelu:
        pre = torch.empty((R, nout), device=x.device, dtype=_f32)
    _lib.call("tmf_tok_linear_fwd", x.data_ptr(), w.data_ptr(), _ptr(bias), _ptr(residual), y.data_ptr(), R, K, nout,
              _ptr(g), _ptr(b), float(eps), _ptr(mean), _ptr(rstd), _ptr(ln_out), _ptr(pre), _stream())
    return y, (mean, rstd, ln_out), pre


def tok_linear_bwd_input(dy, w, gelu_pre=None, ln=None, add1=None, add2=None, ln_partial=None, bias_partial=None,
                         partial_stride=0):
    """dx = E(dy @ w); ln = (x, mean, rstd, gamma) selects the LayerNorm-backward epilogue.  ln_partial /
    bias_partial are (tensor, column offset) pairs into one [row blocks][partial_stride] workspace."""
    R, nout = dy.shape
    K = w.shape[1]
    dx = torch.empty((R, K), device=dy.device, dtype=_f32)
    lx = lm = lr = lg = None
    if ln is not None:
        lx, lm, lr, lg = ln

    def off(pair):
        return None if pair is None else pair[0].data_ptr() + 4 * pair[1]
    _lib.call("tmf_tok_linear_bwd_input", dy.data_ptr(), w.data_ptr(), dx.data_ptr(), R, nout, K, _ptr(gelu_pre),
              _ptr(lx), _ptr(lm), _ptr(lr), _ptr(lg), _ptr(add1), _ptr(add2), off(ln_partial), off(bias_partial),
              partial_stride, _stream())
    return dx


# False / TMF_FUSE_TOKENS=0: every op of the block is its own launch (kept for A/B measurements and tests)
FUSE_TOKEN_LINEARS = os.environ.get("TMF_FUSE_TOKENS", "1") != "0"


def tok_wgrad_multi(pairs):
    """[dy_p^T @ x_p for (dy_p, x_p) in pairs] — all weight gradients of a block in one launch (+ one reduction)."""
    import ctypes as C
    n = len(pairs)
    Rs = (C.c_int * n)(*[dy.shape[0] for dy, _ in pairs])
    Ns = (C.c_int * n)(*[dy.shape[1] for dy, _ in pairs])
    Ks = (C.c_int * n)(*[x.shape[1] for _, x in pairs])
    dev = pairs[0][0].device
    flat = torch.empty(sum(dy.shape[1] * x.shape[1] for dy, x in pairs), device=dev, dtype=_f32)
    outs, o = [], 0
    for dy, x in pairs:
        m = dy.shape[1] * x.shape[1]
        outs.append(flat[o:o + m].view(dy.shape[1], x.shape[1]))
        o += m
    dys = (C.c_void_p * n)(*[dy.data_ptr() for dy, _ in pairs])
    xs = (C.c_void_p * n)(*[x.data_ptr() for _, x in pairs])
    dws = (C.c_void_p * n)(*[t.data_ptr() for t in outs])
    nbytes = _lib.query("tmf_tok_wgrad_multi_workspace_bytes", n, Ns, Ks)
    ws = torch.empty(nbytes // 4, device=dev, dtype=_f32)
    _lib.call("tmf_tok_wgrad_multi", n, dys, xs, dws, Rs, Ns, Ks, ws.data_ptr(), nbytes, _stream())
    return outs


def fused_block_supported(dim, inner, mlp):
    return FUSE_TOKEN_LINEARS and dim == 128 and inner % 128 == 0 and mlp % 128 == 0


class TransformerLayer(torch.autograd.Function):
    """x <- Attention(LayerNorm(x), context) + x ; x <- FeedForward(LayerNorm(x)) + x   as 6 launches forward
    (to_q with LayerNorm prologue, to_kv, attention, to_out + bias + x, Linear + bias + GELU with LayerNorm
    prologue, Linear + bias + x) and 7 + 1 (all five weight gradients) + 2 reductions backward."""

    @staticmethod
    def forward(ctx, x, context, g1, b1n, wq, wkv, wo, bo, g2, b2n, w1, b1, w2, b2, heads, scale, eps1, eps2):
        x, context = _chk(x, "x"), _chk(context, "context")
        B, N, dim = x.shape
        M = context.shape[1]
        inner = wq.shape[0]
        x2d, c2d = x.view(B * N, dim), context.view(B * M, dim)
        q, (mean1, rstd1, a), _ = tok_linear_fwd(x2d, wq, ln=(g1, b1n, eps1), keep_ln_out=True)
        kv, _, _ = tok_linear_fwd(c2d, wkv)
        dh = inner // heads
        out = torch.empty((B * N, inner), device=x.device, dtype=_f32)
        lse = torch.empty((B, heads, N), device=x.device, dtype=_f32)
        _lib.call("tmf_xattn_fwd", q.data_ptr(), kv.data_ptr(), kv.data_ptr() + inner * 4, out.data_ptr(),
                  lse.data_ptr(), B, heads, N, M, dh, inner, 2 * inner, float(scale), _stream())
        x1, _, _ = tok_linear_fwd(out, wo, bias=bo, residual=x2d)
        g, (mean2, rstd2, f), h = tok_linear_fwd(x1, w1, bias=b1, ln=(g2, b2n, eps2), gelu=True, keep_ln_out=True)
        x2, _, _ = tok_linear_fwd(g, w2, bias=b2, residual=x1)
        ctx.save_for_backward(x2d, c2d, g1, wq, wkv, wo, g2, w1, w2, mean1, rstd1, a, q, kv, out, lse, x1, mean2,
                              rstd2, f, h, g)
        ctx.cfg = (B, N, M, dim, inner, heads, float(scale))
        return x2.view(B, N, dim)

    @staticmethod
    def backward(ctx, dx2):
        (x2d, c2d, g1, wq, wkv, wo, g2, w1, w2, mean1, rstd1, a, q, kv, out, lse, x1, mean2, rstd2, f, h,
         g) = ctx.saved_tensors
        B, N, M, dim, inner, heads, scale = ctx.cfg
        mlp = w1.shape[0]
        R = B * N
        dx2 = _chk(dx2, "grad_output").view(R, dim)
        nblk = _lib.query("tmf_tok_row_blocks", R)
        # one [row blocks][stride] workspace for every bias / LayerNorm parameter gradient of the block
        o_b2, o_b1, o_bo, o_ln2, o_ln1 = 0, dim, dim + mlp, 2 * dim + mlp, 4 * dim + mlp
        stride = 6 * dim + mlp
        part = torch.empty((nblk, stride), device=dx2.device, dtype=_f32)
        dh_ = tok_linear_bwd_input(dx2, w2, gelu_pre=h, bias_partial=(part, o_b2), partial_stride=stride)
        dx1 = tok_linear_bwd_input(dh_, w1, ln=(x1, mean2, rstd2, g2), add1=dx2, ln_partial=(part, o_ln2),
                                   bias_partial=(part, o_b1), partial_stride=stride)
        dout = tok_linear_bwd_input(dx1, wo, bias_partial=(part, o_bo), partial_stride=stride)
        dq = torch.empty_like(q)
        dkv = torch.empty_like(kv)
        dhd = inner // heads
        _lib.call("tmf_xattn_bwd", q.data_ptr(), kv.data_ptr(), kv.data_ptr() + inner * 4, out.data_ptr(),
                  lse.data_ptr(), dout.data_ptr(), dq.data_ptr(), dkv.data_ptr(), dkv.data_ptr() + inner * 4,
                  B, heads, N, M, dhd, inner, 2 * inner, 2 * inner, scale, _stream())
        dctx = tok_linear_bwd_input(dkv, wkv) if ctx.needs_input_grad[1] else None
        dx = tok_linear_bwd_input(dq, wq, ln=(x2d, mean1, rstd1, g1), add1=dx1, ln_partial=(part, o_ln1),
                                  partial_stride=stride)
        sums = torch.empty(stride, device=dx2.device, dtype=_f32)
        _lib.call("tmf_colsum_finalize", part.data_ptr(), nblk, stride, sums.data_ptr(), _stream())
        dw2, dw1, dwo, dwkv, dwq = tok_wgrad_multi([(dx2, g), (dh_, f), (dx1, out), (dkv, c2d), (dq, a)])
        return (dx.view(B, N, dim), None if dctx is None else dctx.view(B, M, dim),
                sums[o_ln1:o_ln1 + dim], sums[o_ln1 + dim:o_ln1 + 2 * dim], dwq, dwkv, dwo, sums[o_bo:o_bo + dim],
                sums[o_ln2:o_ln2 + dim], sums[o_ln2 + dim:o_ln2 + 2 * dim], dw1, sums[o_b1:o_b1 + mlp], dw2,
                sums[o_b2:o_b2 + dim], None, None, None, None)


def transformer_layer(x, context, ln1, attn, ln2, ff):
    """ln1 / ln2: nn.LayerNorm; attn: networks.Attention; ff: networks.FeedForward (parameter containers)."""
    return TransformerLayer.apply(x, context, ln1.weight, ln1.bias, attn.to_q.weight, attn.to_kv.weight,
                                  attn.to_out[0].weight, attn.to_out[0].bias, ln2.weight, ln2.bias,
                                  ff.net[0].weight, ff.net[0].bias, ff.net[3].weight, ff.net[3].bias,
                                  attn.heads, attn.scale, ln1.eps, ln2.eps)


# --------------------------------------------------------------------------------------
# whole fusion transformer in one call per pass                      (networks.py:255-281)
# --------------------------------------------------------------------------------------

FUSION_ONE_CALL = os.environ.get("TMF_FUSION_C", "1") != "0"
# the fused per-instance kernels (csrc/xformer_fused.hip) inside the one-call path; "0" keeps one launch per Linear
FUSION_FUSED_KERNELS = os.environ.get("TMF_FUSION_FUSED", "1") != "0"


def fusion_fused_supported(N, dim, heads, dim_head, mlp):
    """Shapes the fused per-instance kernels take (tmf_xf_supported, csrc/xformer_fused.hip)."""
    return (FUSION_ONE_CALL and FUSION_FUSED_KERNELS and dim == 128 and (heads, dim_head) in ((4, 32), (8, 16)) and mlp == 512
            and 1 <= N <= 512)


def dropout_keep_mask(drop, shape, device):
    """Scaled keep-mask of an nn.Dropout in train mode (None when inactive).  A module that provides
    ``tmf_keep_mask(training)`` (tests: fixed masks captured from the reference) supplies its own."""
    if hasattr(drop, "tmf_keep_mask"):
        m = drop.tmf_keep_mask(drop.training)
        return None if m is None else m.to(device=device, dtype=_f32).reshape(shape).contiguous()
    p = float(getattr(drop, "p", 0.0))
    if not drop.training or p <= 0.0:
        return None
    if p >= 1.0:
        return torch.zeros(shape, device=device, dtype=_f32)
    return torch.empty(shape, device=device, dtype=_f32).bernoulli_(1.0 - p).div_(1.0 - p)


_MASK_CALLS = 0          # fallback call counter (a generator without an offset API)


def _mask_stream(device):
    """(seed, offset) of the next tmf_dropout_keep_masks call, taken from torch's OWN device generator — the Philox stream
    position torch's dropout kernels consume — and advanced past it: `torch.manual_seed(s)` therefore reproduces the masks
    of a run exactly as it reproduces nn.Dropout's (seed and offset are host-side integers: no synchronisation)."""
    global _MASK_CALLS
    try:
        gen = torch.cuda.default_generators[device.index if device.index is not None else torch.cuda.current_device()]
        off = gen.get_offset()
        gen.set_offset(off + 4)                              # (torch's offsets move in multiples of 4)
        return gen.initial_seed() & 0xFFFFFFFFFFFFFFFF, off
    except Exception:                                        # pragma: no cover
        _MASK_CALLS += 1
        return torch.initial_seed() & 0xFFFFFFFFFFFFFFFF, (1 << 40) + _MASK_CALLS


def dropout_keep_masks(requests, device):
    """Scaled keep-masks for MANY Dropout modules in ONE launch (tmf_dropout_keep_masks: a counter-based Philox generator
    keyed by the seed and stream offset of torch's device generator, so torch.manual_seed makes a run reproducible): requests = [(drop, shape), ...] -> list
    of fp32 masks (None where the module is inactive).  The two Dropout(0.5) of fc_cls were 6 stock launches per step, the
    18 masks of a depth-3 fusion block with dropout 36; modules that bring their own mask (``tmf_keep_mask``: the tests'
    fixed masks) and p >= 1 are served as in dropout_keep_mask."""
    import ctypes as C
    out = [None] * len(requests)
    live = []
    for i, (drop, shape) in enumerate(requests):
        p = float(getattr(drop, "p", 0.0))
        if hasattr(drop, "tmf_keep_mask") or p >= 1.0 or p <= 0.0 or not drop.training:
            out[i] = dropout_keep_mask(drop, shape, device)
        else:
            live.append((i, 1.0 - p, int(torch.Size(shape).numel())))
    for s0 in range(0, len(live), _lib.MASK_SEGMENTS):
        part = live[s0:s0 + _lib.MASK_SEGMENTS]
        sizes = [(n + 3) & ~3 for _i, _k, n in part]                     # 16-byte aligned segments
        flat = torch.empty(sum(sizes), device=device, dtype=_f32)
        ptrs, off = [], 0
        for (i, _k, n), sz in zip(part, sizes):
            out[i] = flat[off:off + n].view(requests[i][1])
            ptrs.append(flat.data_ptr() + 4 * off)
            off += sz
        seed, offset = _mask_stream(device)
        _lib.call("tmf_dropout_keep_masks", len(part), (C.c_void_p * len(part))(*ptrs), (C.c_long * len(part))(*[n for _i, _k, n in part]),
                  (C.c_float * len(part))(*[k for _i, k, _n in part]), seed, offset, _stream())
    return out


def fusion_one_call_supported(dim, inner, mlp, dim_head, depth):
    return (FUSION_ONE_CALL and FUSE_TOKEN_LINEARS and dim == 128 and inner % 128 == 0 and mlp % 128 == 0
            and dim_head in (8, 16, 32, 64) and 0 < depth <= 16)


class FusionTrain(torch.autograd.Function):
    """CrossTransformer_MOD_AVG forward / backward as ONE library call each (tmf_fusion_train_fwd / _bwd,
    csrc/fusion_path.hip).  forward(mri_tok, pet_tok, cfg, *params): params = per Transformer instance (mri enc of layer
    0, pet enc of layer 0, mri enc of layer 1, ...) the 14 tensors of _lib.XFORMER_PTRS order -> cls (B, 4*dim)."""

    @staticmethod
    def forward(ctx, mri, pet, cfg, *params):
        import ctypes as C
        mri, pet = _chk(mri, "mri_tokens"), _chk(pet, "pet_tokens")
        heads, dim_head, mlp, depth, eps = cfg[:5]
        drops = cfg[5] if len(cfg) > 5 else None      # per instance the three nn.Dropout modules (to_out, GELU, Linear 2)
        B, N, dim = mri.shape
        if pet.shape != mri.shape:
            raise _lib.TmfError(f"token shapes differ: {tuple(mri.shape)} vs {tuple(pet.shape)}")
        desc = _lib.FusionDesc(B=B, N=N, dim=dim, heads=heads, dim_head=dim_head, mlp=mlp, depth=depth,
                               flags=0 if FUSION_FUSED_KERNELS else _lib.FUSION_PER_OP)
        inst = (_lib.XformerParams * (2 * depth))()
        masks = []
        for i in range(2 * depth):
            for j, name in enumerate(_lib.XFORMER_PTRS):
                t = params[14 * i + j]
                if not (t.is_cuda and t.dtype == _f32 and t.is_contiguous()):
                    raise _lib.TmfError(f"Transformer instance {i}: {name} must be a contiguous float32 HIP tensor")
                setattr(inst[i], name, t.data_ptr())
            inst[i].eps1, inst[i].eps2, inst[i].epsf = eps[i]
        if drops is not None:                    # every keep-mask of the step in one draw per distinct p
            req = [(drop, (B * N, width)) for i in range(2 * depth) for drop, width in zip(drops[i], (dim, mlp, dim))]
            for j, mk in enumerate(dropout_keep_masks(req, mri.device)):
                if mk is not None:
                    masks.append(mk)
                    setattr(inst[j // 3], ("mask_o", "mask_g", "mask_f")[j % 3], mk.data_ptr())
        nsaved = _lib.query("tmf_fusion_saved_bytes", C.byref(desc))
        if nsaved == 0:
            raise _lib.TmfError("tmf_fusion_saved_bytes: " + (_lib.load().tmf_last_error_string() or b"").decode())
        saved = torch.empty(nsaved, device=mri.device, dtype=torch.uint8)
        cls = torch.empty((B, 4 * dim), device=mri.device, dtype=_f32)
        _lib.call("tmf_fusion_train_fwd", C.byref(desc), mri.data_ptr(), pet.data_ptr(), inst, saved.data_ptr(), nsaved,
                  cls.data_ptr(), _stream())
        ctx.save_for_backward(mri, pet, saved, *params)     # parameters too: autograd then rejects an in-place update
        ctx.desc, ctx.inst = desc, inst                     # between forward and backward; their pointers stay valid:
        ctx.masks = masks                                   # the ctypes structs are reused as they are
        ctx.param_ptrs = [t.data_ptr() for t in params]
        # Everything backward needs besides the incoming gradient is set up NOW, while the GPU is busy with the forward:
        # backward starts right behind the reference step's two loss.item() host syncs (kfold_train_adversarial.py:127-128),
        # where every microsecond of host preparation is a microsecond of idle GPU (measured: 140 us of gap in front of the
        # first fusion-backward kernel).
        ctx.bwd = FusionTrain._prepare_backward(desc, mri, pet) if any(ctx.needs_input_grad) else None
        return cls

    @staticmethod
    def _prepare_backward(desc, mri, pet):
        import ctypes as C
        depth, dim, mlp = desc.depth, desc.dim, desc.mlp
        inner = desc.heads * desc.dim_head
        n_inst = 2 * depth
        grads = (_lib.XformerGrads * n_inst)()
        # ONE flat gradient buffer for the whole fusion, cut by ONE split call; per instance
        # [b2 | b1 | bo | ln2 g | ln2 b | ln1 g | ln1 b | lnf g | lnf b | dwq | dwkv | dwo | dw1 | dw2]
        # (the first seven are the contiguous `small` region of tmf_xformer_grads, the next two its `lnf`)
        sizes = [dim, mlp, dim, dim, dim, dim, dim, dim, dim, inner * dim, 2 * inner * dim, dim * inner, mlp * dim, dim * mlp]
        per = sum(sizes)
        flat = torch.empty(n_inst * per, device=mri.device, dtype=_f32)
        parts = flat.split(sizes * n_inst)
        base = flat.data_ptr()
        o_lnf = 4 * (6 * dim + mlp)
        o_w = [o_lnf + 4 * 2 * dim]
        for n in sizes[9:13]:
            o_w.append(o_w[-1] + 4 * n)
        out = [None, None, None]
        for i in range(n_inst):
            g = grads[i]
            b0 = base + 4 * i * per
            g.small, g.lnf = b0, b0 + o_lnf
            g.dwq, g.dwkv, g.dwo, g.dw1, g.dw2 = (b0 + o for o in o_w)
            b2, b1, bo, g2, be2, g1, be1, gf, bf, dwq, dwkv, dwo, dw1, dw2 = parts[14 * i:14 * i + 14]
            # order of _lib.XFORMER_PTRS
            out += [g1, be1, dwq.view(inner, dim), dwkv.view(2 * inner, dim), dwo.view(dim, inner), bo, g2, be2,
                    dw1.view(mlp, dim), b1, dw2.view(dim, mlp), b2, gf, bf]
        dm = torch.empty_like(mri)
        dp = torch.empty_like(pet)
        nscr = _lib.query("tmf_fusion_bwd_scratch_bytes", C.byref(desc))
        scratch = torch.empty(nscr, device=mri.device, dtype=torch.uint8)
        out[0], out[1] = dm, dp
        return grads, flat, out, dm, dp, scratch, nscr

    @staticmethod
    def backward(ctx, dcls):
        import ctypes as C
        mri, pet, saved = ctx.saved_tensors[:3]
        desc, inst = ctx.desc, ctx.inst
        dcls = _chk(dcls, "grad_output")
        prep = ctx.bwd if ctx.bwd is not None else FusionTrain._prepare_backward(desc, mri, pet)
        ctx.bwd = None
        grads, flat, out, dm, dp, scratch, nscr = prep
        _lib.call("tmf_fusion_train_bwd", C.byref(desc), mri.data_ptr(), pet.data_ptr(), inst, saved.data_ptr(),
                  saved.numel(), dcls.data_ptr(), grads, dm.data_ptr(), dp.data_ptr(), scratch.data_ptr(), nscr, _stream())
        # (Round 4 measured the parameter-gradient launches of this pass — column sums, the weight-gradient launch: 0.08 ms
        #  of latency-bound work nothing in backward waits for — on a side stream beside the encoders' backward: per-step
        #  medians 14.54 / 14.60 / 14.60 ms against 14.64 / 14.56 / 14.55, and 7.41 / 7.44 / 7.41 against 7.42 / 7.44 / 7.39 at
        #  128^3 bf16: level — what leaves the serial token phase is paid back as CU time beside the encoders' kernels.  Dropped.
        #  A first, flawed version was 2-4 % SLOWER: its end-of-backward callback kept the gradient VIEWS alive, and autograd
        #  adopts a view as .grad only while it holds the last reference — a kept view is cloned: 84 copy launches.)
        if _FLAT_GRAD_CONSUMERS and all(ctx.needs_input_grad[3:]):
            _publish_flat_grads(flat, ctx.param_ptrs, out[3:], [(0, flat.numel(), None, "next")])
        return tuple(out)


# --------------------------------------------------------------------------------------
# the dense heads of model_ad in one launch per direction            (mymodel.py:190-194, 209-221)
# --------------------------------------------------------------------------------------

HEADS_ONE_CALL = os.environ.get("TMF_HEADS_C", "1") != "0"


class HeadsAD(torch.autograd.Function):
    """fc_cls(cls) and D(revgrad(mean_n mri_tok)), D(revgrad(mean_n pet_tok)) as ONE kernel forward and ONE backward
    (tmf_heads_fwd / _bwd, csrc/heads.hip).  forward(cls, mri_tok, pet_tok, mask1, mask2, cfg, buffers, *params) with
    params in _lib.HEADS_PARAMS order, buffers in _lib.HEADS_BUFFERS order (None = no running statistics), masks = scaled
    Dropout keep-masks or None, cfg = (training, (momentum x3), (eps x3), revgrad alpha) -> (logits, D_MRI, D_PET)."""

    @staticmethod
    def forward(ctx, cls, mri, pet, mask1, mask2, cfg, buffers, *params):
        import ctypes as C
        cls, mri, pet = _chk(cls, "cls"), _chk(mri, "mri_tokens"), _chk(pet, "pet_tokens")
        training, momentum, eps, alpha = cfg
        B, N, dim = mri.shape
        P = dict(zip(_lib.HEADS_PARAMS, params))
        desc = _lib.HeadsDesc(B=B, N=N, dim=dim, H1=P["fc0_w"].shape[0], H2=P["fc4_w"].shape[0], HD=P["d0_w"].shape[0],
                              NC=P["fc8_w"].shape[0], training=int(training))
        for i in range(3):
            desc.momentum[i], desc.eps[i] = momentum[i], eps[i]
        prm = _lib.HeadsParams()
        for name, t in P.items():
            if not (t.is_cuda and t.dtype == _f32 and t.is_contiguous()):
                raise _lib.TmfError(f"heads: {name} must be a contiguous float32 tensor on the HIP device")
            setattr(prm, name, t.data_ptr())
        for name, t in zip(_lib.HEADS_BUFFERS, buffers):
            setattr(prm, name, _ptr(t))
        m1 = None if mask1 is None else _chk(mask1, "mask1")
        m2 = None if mask2 is None else _chk(mask2, "mask2")
        nsaved = _lib.query("tmf_heads_saved_bytes", C.byref(desc))
        if nsaved == 0:
            raise _lib.TmfError("tmf_heads_saved_bytes: " + (_lib.load().tmf_last_error_string() or b"").decode())
        saved = torch.empty(nsaved // 4, device=cls.device, dtype=_f32)
        outs = torch.empty((3, B, desc.NC), device=cls.device, dtype=_f32)
        _lib.call("tmf_heads_fwd", C.byref(desc), cls.data_ptr(), mri.data_ptr(), pet.data_ptr(), _ptr(m1), _ptr(m2),
                  C.byref(prm), outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), saved.data_ptr(), nsaved, _stream())
        ctx.save_for_backward(cls, saved, *params)
        ctx.masks = (m1, m2)
        ctx.desc, ctx.prm, ctx.alpha = desc, prm, float(alpha)
        ctx.tok_shape = tuple(mri.shape)
        ctx.param_ptrs = [t.data_ptr() for t in params]
        # backward's buffers and gradient table, set up while the GPU is still busy with the forward (see FusionTrain)
        ctx.bwd = HeadsAD._prepare_backward(desc, cls, params, ctx.tok_shape) if any(ctx.needs_input_grad) else None
        return outs[0], outs[1], outs[2]

    @staticmethod
    def _prepare_backward(desc, cls, params, tok_shape):
        import ctypes as C
        dev = cls.device
        sizes = [p.numel() for p in params]
        flat = torch.empty(sum(sizes), device=dev, dtype=_f32)
        parts = flat.split(sizes)
        g = _lib.HeadsGrads()
        o = flat.data_ptr()
        for name, n in zip(_lib.HEADS_PARAMS, sizes):
            setattr(g, name, o)
            o += 4 * n
        d_cls = torch.empty_like(cls)
        d_tok = torch.empty((2,) + tok_shape, device=dev, dtype=_f32)
        nscr = _lib.query("tmf_heads_bwd_scratch_bytes", C.byref(desc))
        scratch = torch.empty(nscr // 4, device=dev, dtype=_f32)
        zero = torch.zeros((desc.B, desc.NC), device=dev, dtype=_f32)      # stands in for an output nobody differentiated
        grads = [t if p.dim() == 1 else t.view(p.shape) for t, p in zip(parts, params)]
        return g, flat, grads, d_cls, d_tok, scratch, nscr, zero

    @staticmethod
    def backward(ctx, dlo, ddm, ddp):
        import ctypes as C
        cls, saved = ctx.saved_tensors[:2]
        params = ctx.saved_tensors[2:]
        desc = ctx.desc
        prep = ctx.bwd if ctx.bwd is not None else HeadsAD._prepare_backward(desc, cls, params, ctx.tok_shape)
        ctx.bwd = None
        g, flat, grads, d_cls, d_tok, scratch, nscr, zero = prep
        # the three output gradients go to the library as they are (no stack / copy launch in front of the kernel)
        dl = [zero if t is None else (t if (t.dtype == _f32 and t.is_contiguous()) else t.to(_f32).contiguous())
              for t in (dlo, ddm, ddp)]
        m1, m2 = ctx.masks
        _lib.call("tmf_heads_bwd", C.byref(desc), cls.data_ptr(), _ptr(m1), _ptr(m2), C.byref(ctx.prm), saved.data_ptr(),
                  saved.numel() * 4, dl[0].data_ptr(), dl[1].data_ptr(), dl[2].data_ptr(), C.byref(g), d_cls.data_ptr(),
                  d_tok[0].data_ptr(), d_tok[1].data_ptr(), ctx.alpha, scratch.data_ptr(), nscr, _stream())
        if _FLAT_GRAD_CONSUMERS and all(ctx.needs_input_grad[7:]):
            _publish_flat_grads(flat, ctx.param_ptrs, grads, [(0, flat.numel(), None, "next")])
        return (d_cls, d_tok[0], d_tok[1], None, None, None, None, *grads)


class HeadsCNN(torch.autograd.Function):
    """The heads of the CNN-only models as ONE kernel per direction (tmf_heads_cnn_fwd / _bwd, csrc/heads.hip):
    model_CNN_ad: fc_cls(cat[mean_n mri, mean_n pet]), D(revgrad(mean_n mri)), D(revgrad(mean_n pet))   (mymodel.py:164-178)
    model_single: fc(mean_n tok)                                                                          (mymodel.py:33-39).
    forward(mri_tok, pet_tok | None, cfg, buffers, *params): params = fc0_w, fc0_b, fc2_w, fc2_b [, d0_w, d0_b, dbn_g, dbn_b,
    d3_w, d3_b]; buffers = (running_mean, running_var) of D's BatchNorm1d or (None, None); cfg = (training, momentum, eps,
    revgrad alpha) -> (logits, D_MRI, D_PET) or (logits,)."""

    @staticmethod
    def forward(ctx, mri, pet, cfg, buffers, *params):
        import ctypes as C
        mri = _chk(mri, "mri_tokens")
        M = 1 if pet is None else 2
        if M == 2:
            pet = _chk(pet, "pet_tokens")
            if pet.shape != mri.shape:
                raise _lib.TmfError(f"token shapes differ: {tuple(mri.shape)} vs {tuple(pet.shape)}")
        with_d = len(params) == len(_lib.HEADS_CNN_PARAMS)
        if not with_d and len(params) != 4:
            raise _lib.TmfError(f"heads_cnn: {len(params)} parameter tensors (4 without, 10 with the discriminator)")
        training, momentum, eps, alpha = cfg
        B, N, dim = mri.shape
        P = dict(zip(_lib.HEADS_CNN_PARAMS, params))
        desc = _lib.HeadsCnnDesc(B=B, N=N, dim=dim, M=M, H=P["fc0_w"].shape[0], HD=P["d0_w"].shape[0] if with_d else 0,
                                 NC=P["fc2_w"].shape[0], training=int(training), momentum=momentum, eps=eps)
        prm = _lib.HeadsCnnParams()
        for name, t in P.items():
            if not (t.is_cuda and t.dtype == _f32 and t.is_contiguous()):
                raise _lib.TmfError(f"heads_cnn: {name} must be a contiguous float32 tensor on the HIP device")
            setattr(prm, name, t.data_ptr())
        prm.dbn_rm, prm.dbn_rv = _ptr(buffers[0]), _ptr(buffers[1])
        nsaved = _lib.query("tmf_heads_cnn_saved_bytes", C.byref(desc))
        if nsaved == 0:
            raise _lib.TmfError("tmf_heads_cnn_saved_bytes: " + (_lib.load().tmf_last_error_string() or b"").decode())
        saved = torch.empty(nsaved // 4, device=mri.device, dtype=_f32)
        outs = torch.empty((3 if with_d else 1, B, desc.NC), device=mri.device, dtype=_f32)
        _lib.call("tmf_heads_cnn_fwd", C.byref(desc), mri.data_ptr(), _ptr(pet), C.byref(prm), outs[0].data_ptr(),
                  outs[1].data_ptr() if with_d else None, outs[2].data_ptr() if with_d else None, saved.data_ptr(), nsaved,
                  _stream())
        ctx.save_for_backward(saved, *params)
        ctx.desc, ctx.prm, ctx.alpha, ctx.with_d = desc, prm, float(alpha), with_d
        ctx.tok_shape = tuple(mri.shape)
        ctx.param_ptrs = [t.data_ptr() for t in params]
        ctx.bwd = HeadsCNN._prepare_backward(desc, mri.device, params, ctx.tok_shape) if any(ctx.needs_input_grad) else None
        return (outs[0], outs[1], outs[2]) if with_d else outs[0]

    @staticmethod
    def _prepare_backward(desc, dev, params, tok_shape):
        import ctypes as C
        sizes = [p.numel() for p in params]
        flat = torch.empty(sum(sizes), device=dev, dtype=_f32)
        parts = flat.split(sizes)
        g = _lib.HeadsCnnGrads()
        o = flat.data_ptr()
        for name, n in zip(_lib.HEADS_CNN_PARAMS, sizes):
            setattr(g, name, o)
            o += 4 * n
        d_tok = torch.empty((desc.M,) + tok_shape, device=dev, dtype=_f32)
        nscr = _lib.query("tmf_heads_cnn_bwd_scratch_bytes", C.byref(desc))
        scratch = torch.empty(nscr // 4, device=dev, dtype=_f32)
        zero = torch.zeros((desc.B, desc.NC), device=dev, dtype=_f32)      # stands in for an output nobody differentiated
        grads = [t if p.dim() == 1 else t.view(p.shape) for t, p in zip(parts, params)]
        return g, flat, grads, d_tok, scratch, nscr, zero

    @staticmethod
    def backward(ctx, dlo, ddm=None, ddp=None):
        import ctypes as C
        saved = ctx.saved_tensors[0]
        params = ctx.saved_tensors[1:]
        desc = ctx.desc
        prep = ctx.bwd if ctx.bwd is not None else HeadsCNN._prepare_backward(desc, saved.device, params, ctx.tok_shape)
        ctx.bwd = None
        g, flat, grads, d_tok, scratch, nscr, zero = prep
        dl = [zero if t is None else (t if (t.dtype == _f32 and t.is_contiguous()) else t.to(_f32).contiguous())
              for t in (dlo, ddm, ddp)]
        two = desc.M == 2
        _lib.call("tmf_heads_cnn_bwd", C.byref(desc), C.byref(ctx.prm), saved.data_ptr(), saved.numel() * 4, dl[0].data_ptr(),
                  dl[1].data_ptr() if ctx.with_d else None, dl[2].data_ptr() if ctx.with_d else None, C.byref(g),
                  d_tok[0].data_ptr(), d_tok[1].data_ptr() if two else None, ctx.alpha, scratch.data_ptr(), nscr, _stream())
        if _FLAT_GRAD_CONSUMERS and all(ctx.needs_input_grad[4:]):
            _publish_flat_grads(flat, ctx.param_ptrs, grads, [(0, flat.numel(), None, "next")])
        return (d_tok[0], d_tok[1] if two else None, None, None, *grads)


# --------------------------------------------------------------------------------------
# token pooling head                                                (networks.py:276-281)
# --------------------------------------------------------------------------------------

class TokenPool(torch.autograd.Function):
    """cat[mean_n mri, mean_n pet, max_n mri, max_n pet] -> (B, 4*dim)."""

    @staticmethod
    def forward(ctx, mri, pet):
        mri, pet = _chk(mri, "mri_tokens"), _chk(pet, "pet_tokens")
        B, N, dim = mri.shape
        if pet.shape != mri.shape:
            raise _lib.TmfError(f"token shapes differ: {tuple(mri.shape)} vs {tuple(pet.shape)}")
        cls = torch.empty((B, 4 * dim), device=mri.device, dtype=_f32)
        arg = torch.empty((B, 2, dim), device=mri.device, dtype=torch.int32)
        _lib.call("tmf_token_pool_fwd", mri.data_ptr(), pet.data_ptr(), cls.data_ptr(), arg.data_ptr(),
                  B, N, dim, _stream())
        ctx.save_for_backward(arg)
        ctx.shape = (B, N, dim)
        return cls

    @staticmethod
    def backward(ctx, dcls):
        (arg,) = ctx.saved_tensors
        B, N, dim = ctx.shape
        dcls = _chk(dcls, "grad_output")
        dm = torch.empty((B, N, dim), device=dcls.device, dtype=_f32)
        dp = torch.empty((B, N, dim), device=dcls.device, dtype=_f32)
        _lib.call("tmf_token_pool_bwd", dcls.data_ptr(), arg.data_ptr(), dm.data_ptr(), dp.data_ptr(),
                  B, N, dim, _stream())
        return dm, dp


def token_pool(mri, pet):
    return TokenPool.apply(mri, pet)
