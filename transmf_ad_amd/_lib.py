"""ctypes binding of libtmf_hip.so (C ABI: include/tmf_hip.h).

There is NO fallback: if the library is missing, or a call returns non-zero, this
module raises.  The product path never routes around the HIP kernels.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtmf_hip.so")

POOL_NONE, POOL_MAX2, POOL_AVG2 = 0, 1, 2
DW_TAPMAJOR, DW_REFERENCE = 0, 1
_POOL = {None: POOL_NONE, "none": POOL_NONE, "max": POOL_MAX2, "avg": POOL_AVG2}

_p = C.c_void_p
_i = C.c_int
_f = C.c_float
_l = C.c_long
_d = C.c_double
_z = C.c_size_t

# name -> (restype, argtypes); mirrors include/tmf_hip.h declaration by declaration
PROTOTYPES = {
    "tmf_version": (_i, []),
    "tmf_last_error_string": (C.c_char_p, []),
    "tmf_set_option": (_i, [C.c_char_p, _i]),
    "tmf_conv3d_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "tmf_conv3d_fwd_affine": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _f, _p]),
    "tmf_conv3d_stat_blocks": (_i, [_i, _i, _i, _i, _i, _i, _i]),
    "tmf_conv3d_fwd_kernel_name": (C.c_char_p, [_i, _i, _i, _i, _i, _i, _i]),
    "tmf_conv3d_wgrad_kernel_name": (C.c_char_p, [_i, _i, _i, _i, _i, _i, _i]),
    "tmf_conv3d_wgrad_workspace_bytes": (_z, [_i, _i, _i, _i, _i, _i, _i]),
    "tmf_conv3d_wgrad": (_i, [_p, _p, _p, _p, _z, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "tmf_conv3d_fwd_bf16": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "tmf_conv3d_wgrad_bf16_workspace_bytes": (_z, [_i, _i, _i, _i, _i, _i]),
    "tmf_conv3d_wgrad_bf16": (_i, [_p, _p, _p, _p, _z, _i, _i, _i, _i, _i, _i, _p]),
    "tmf_conv3d_fwd_split": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "tmf_conv3d_bf16_stat_blocks": (_i, [_i, _i, _i, _i]),
    "tmf_conv3d_split_stat_blocks": (_i, [_i, _i, _i, _i]),
    "tmf_conv3d_fwd_wino": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "tmf_conv3d_wino_ok": (_i, [_i, _i]),
    "tmf_conv3d_fwd_wino_affine": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _p]),
    "tmf_conv3d_wino_stat_blocks": (_i, [_i, _i, _i, _i]),
    "tmf_conv3d_wino_bricks": (_i, [_i, _i, _i, _i]),
    "tmf_conv3d_wino_bricks2": (_i, [_i, _i, _i, _i, _i, _i]),
    "tmf_conv3d_wino_kernel_name": (C.c_char_p, [_i, _i, _i, _i, _i]),
    "tmf_conv3d_wgrad_wino_kernel_name": (C.c_char_p, [_i, _i, _i, _i, _i, _i]),
    "tmf_conv3d_wino_kernel_name2": (C.c_char_p, [_i, _i, _i, _i, _i, _i, _i]),
    "tmf_conv3d_wgrad_wino_tiles": (C.c_long, [_i, _i, _i, _i, _i, _i]),
    "tmf_conv3d_wino_weight_bytes": (_z, [_i, _i]),
    "tmf_wino_x_mode": (_i, []),
    "tmf_c1_split_mode": (_i, []),
    "tmf_snet_algo_flags": (_i, []),
    "tmf_conv_wino_mode": (_i, []),
    "tmf_wino_p_mode": (_i, []),
    "tmf_conv3d_wgrad_wino_ok": (_i, [_i, _i]),
    "tmf_conv3d_wgrad_wino_workspace_bytes": (_z, [_i, _i, _i, _i, _i, _i]),
    "tmf_conv3d_wgrad_wino": (_i, [_p, _p, _p, _p, _z, _i, _i, _i, _i, _i, _i, _i, _p]),
    "tmf_conv3d_fwd_bf16_kernel_name": (C.c_char_p, [_i, _i, _i, _i, _i, _i, _i]),
    "tmf_conv3d_wgrad_bf16_kernel_name": (C.c_char_p, [_i, _i, _i, _i, _i, _i, _i]),
    "tmf_conv3d_c1_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "tmf_conv3d_c1_stat_blocks": (_i, [_i, _i, _i, _i, _i]),
    "tmf_conv3d_c1_wgrad_workspace_bytes": (_z, [_i, _i, _i, _i, _i]),
    "tmf_conv3d_c1_wgrad": (_i, [_p, _p, _p, _p, _z, _i, _i, _i, _i, _i, _i, _p]),
    "tmf_c1_blocks": (_i, [_i, _i, _i, _i, _i]),
    "tmf_c1_stats": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "tmf_c1_gram_bytes": (_z, [_i, _i, _i, _i, _i]),
    "tmf_c1_gram_bytes_bf16": (_z, [_i, _i, _i, _i, _i]),
    "tmf_c1_stats_g": (_i, [_p, _p, _p, _p, _z, _i, _i, _i, _i, _i, _p]),
    "tmf_c1_bwd_fused_workspace_bytes": (_z, [_i, _i, _i, _i, _i]),
    "tmf_c1_bwd_fused": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _z, _i, _i, _i, _i, _i, _f, _i, _p]),
    "tmf_c1_stats_g_bf16": (_i, [_p, _p, _p, _p, _z, _i, _i, _i, _i, _i, _p]),
    "tmf_c1_bwd_fused_bf16": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _z, _i, _i, _i, _i, _i, _f, _i, _i, _p]),
    "tmf_c1_bn_pool_fwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _p]),
    "tmf_c1_bwd_reduce": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _p]),
    "tmf_c1_stats_bf16": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "tmf_c1_bn_pool_fwd_bf16": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _i, _p]),
    "tmf_c1_bwd_reduce_bf16": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _i, _p]),
    "tmf_c1_bwd_wgrad_bf16": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _z, _i, _i, _i, _i, _i, _f, _i, _i, _p]),
    "tmf_conv3d_fwd_bf16_t": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "tmf_conv3d_wgrad_bf16_t": (_i, [_p, _p, _p, _p, _z, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "tmf_bn_act_pool_fwd_t": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _i, _p]),
    "tmf_bn_act_pool_bwd_reduce_t": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _i, _p]),
    "tmf_bn_act_pool_bwd_apply_t": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _i, _p]),
    "tmf_c1_bwd_wgrad_workspace_bytes": (_z, [_i, _i, _i, _i, _i]),
    "tmf_c1_bwd_wgrad": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _z, _i, _i, _i, _i, _i, _f, _i, _p]),
    "tmf_bn_finalize": (_i, [_p, _i, _i, _d, _p, _p, _p, _p, _p, _f, _f, _p, _p, _p, _p, _p]),
    "tmf_bn_eval_coeffs": (_i, [_p, _p, _p, _p, _p, _f, _i, _p, _p, _p]),
    "tmf_bn_act_pool_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _p]),
    "tmf_bn_act_pool_bwd_blocks": (_i, [_i, _i, _i, _i, _i, _i]),
    "tmf_bn_act_pool_bwd_reduce": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _p]),
    "tmf_bn_bwd_finalize": (_i, [_p, _i, _i, _d, _p, _p, _p, _p]),
    "tmf_bn_act_pool_bwd_apply": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _p]),
    "tmf_colsum_finalize": (_i, [_p, _i, _i, _p, _p]),
    "tmf_xattn_fwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _p]),
    "tmf_xattn_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _f, _p]),
    "tmf_layernorm_fwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _f, _p]),
    "tmf_pack_conv_weights": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "tmf_pack_conv_weights_bf16": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "tmf_pack_conv_weights_split3": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "tmf_pack_conv_weights_wino": (_i, [_p, _p, _p, _i, _i, _p]),
    "tmf_pack_conv_weights_wino_multi": (_i, [_i, _p, _p, _p, _p, _p, _p]),
    "tmf_layout_ncdhw_to_ndhwc": (_i, [_p, _p, _i, _i, _l, _p]),
    "tmf_layout_ndhwc_to_ncdhw": (_i, [_p, _p, _i, _i, _l, _p]),
    "tmf_tok_row_blocks": (_i, [_i]),
    "tmf_tok_wgrad_multi_workspace_bytes": (_z, [_i, _p, _p]),
    "tmf_tok_wgrad_multi": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _z, _p]),
    "tmf_tok_linear_fwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _p, _p, _f, _p, _p, _p, _p, _p]),
    "tmf_tok_linear_bwd_input": (_i, [_p, _p, _p, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p]),
    "tmf_layernorm_bwd_blocks": (_i, [_i, _i]),
    "tmf_layernorm_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _p]),
    "tmf_token_pool_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _p]),
    "tmf_token_pool_bwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _p]),
}



class SnetDesc(C.Structure):
    """tmf_snet_desc (include/tmf_hip.h)"""
    _fields_ = [("B", _i), ("D", _i), ("H", _i), ("W", _i), ("dim", _i), ("precision", _i), ("storage_bf16", _i),
                ("momentum", _f * 7), ("eps", _f * 7), ("slope", _f * 7), ("flags", _i)]
SNET_ALONE = 1


class SnetParams(C.Structure):
    _fields_ = [("weight", _p * 7), ("bias", _p * 7), ("gamma", _p * 7), ("beta", _p * 7),
                ("running_mean", _p * 7), ("running_var", _p * 7)]


class SnetGrads(C.Structure):
    _fields_ = [("dweight", _p * 7), ("dbias", _p * 7), ("dgamma", _p * 7), ("dbeta", _p * 7), ("deep_event", _p)]
SNET_DEEP_FROM = 3
MASK_SEGMENTS = 20           # TMF_MASK_SEGMENTS


PROTOTYPES.update({
    "tmf_scale_intensity_workspace_bytes": (_z, [_i]),
    "tmf_volume_minmax": (_i, [_p, _p, _p, _z, _i, _l, _p]),
    "tmf_scale_flip": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "tmf_rotate_x": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "tmf_zoom_area": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "tmf_snet_saved_bytes": (_z, [C.POINTER(SnetDesc)]),
    "tmf_snet_bwd_scratch_bytes": (_z, [C.POINTER(SnetDesc)]),
    "tmf_snet_train_fwd": (_i, [C.POINTER(SnetDesc), _p, C.POINTER(SnetParams), _p, _z, _p, _p]),
    "tmf_snet_train_bwd": (_i, [C.POINTER(SnetDesc), _p, _p, _z, _p, C.POINTER(SnetGrads), _p, _z, _p]),
    "tmf_snet_eval_workspace_bytes": (_z, [C.POINTER(SnetDesc)]),
    "tmf_snet_eval_fwd": (_i, [C.POINTER(SnetDesc), _p, C.POINTER(SnetParams), _p, _z, _p, _p]),
})



class FusionDesc(C.Structure):
    _fields_ = [(n, _i) for n in ("B", "N", "dim", "heads", "dim_head", "mlp", "depth", "flags")]


FUSION_PER_OP = 1


XFORMER_PTRS = ("ln1_g", "ln1_b", "wq", "wkv", "wo", "bo", "ln2_g", "ln2_b", "w1", "b1", "w2", "b2", "lnf_g", "lnf_b")


class XformerParams(C.Structure):
    _fields_ = ([(n, _p) for n in XFORMER_PTRS] + [("eps1", _f), ("eps2", _f), ("epsf", _f)]
                + [(n, _p) for n in ("mask_o", "mask_g", "mask_f")])


class XformerGrads(C.Structure):
    _fields_ = [(n, _p) for n in ("small", "lnf", "dwq", "dwkv", "dwo", "dw1", "dw2")]


PROTOTYPES.update({
    "tmf_debug_xf_trace": (None, [_p, _p, _p]),
    "tmf_fusion_saved_bytes": (_z, [C.POINTER(FusionDesc)]),
    "tmf_fusion_uses_fused": (_i, [C.POINTER(FusionDesc)]),
    "tmf_fusion_bwd_scratch_bytes": (_z, [C.POINTER(FusionDesc)]),
    "tmf_fusion_train_fwd": (_i, [C.POINTER(FusionDesc), _p, _p, C.POINTER(XformerParams), _p, _z, _p, _p]),
    "tmf_fusion_train_bwd": (_i, [C.POINTER(FusionDesc), _p, _p, C.POINTER(XformerParams), _p, _z, _p,
                                  C.POINTER(XformerGrads), _p, _p, _p, _z, _p]),
})



class HeadsDesc(C.Structure):
    _fields_ = [(n, _i) for n in ("B", "N", "dim", "H1", "H2", "HD", "NC", "training")] + [("momentum", _f * 3), ("eps", _f * 3)]


HEADS_PARAMS = ("fc0_w", "fc0_b", "bn1_g", "bn1_b", "fc4_w", "fc4_b", "bn5_g", "bn5_b", "fc8_w", "fc8_b",
                "d0_w", "d0_b", "dbn_g", "dbn_b", "d3_w", "d3_b")
HEADS_BUFFERS = ("bn1_rm", "bn1_rv", "bn5_rm", "bn5_rv", "dbn_rm", "dbn_rv")


class HeadsParams(C.Structure):
    _fields_ = [(n, _p) for n in HEADS_PARAMS + HEADS_BUFFERS]


class HeadsGrads(C.Structure):
    _fields_ = [(n, _p) for n in HEADS_PARAMS]


PROTOTYPES.update({
    "tmf_heads_saved_bytes": (_z, [C.POINTER(HeadsDesc)]),
    "tmf_heads_bwd_scratch_bytes": (_z, [C.POINTER(HeadsDesc)]),
    "tmf_dropout_keep_masks": (_i, [_i, _p, _p, _p, C.c_ulonglong, C.c_ulonglong, _p]),
    "tmf_heads_fwd": (_i, [C.POINTER(HeadsDesc), _p, _p, _p, _p, _p, C.POINTER(HeadsParams), _p, _p, _p, _p, _z, _p]),
    "tmf_heads_bwd": (_i, [C.POINTER(HeadsDesc), _p, _p, _p, C.POINTER(HeadsParams), _p, _z, _p, _p, _p,
                           C.POINTER(HeadsGrads), _p, _p, _p, _f, _p, _z, _p]),
    "tmf_adam_state_elems": (_l, [_i, C.POINTER(_l)]),
    "tmf_adam_step": (_i, [_i, C.POINTER(_p), C.POINTER(_p), C.POINTER(_l), _p, _p, _d, _d, _d, _d, _d, _i, _p]),
})
ADAM_MAX_TENSORS = 160


class HeadsCnnDesc(C.Structure):
    _fields_ = [(n, _i) for n in ("B", "N", "dim", "M", "H", "HD", "NC", "training")] + [("momentum", _f), ("eps", _f)]


HEADS_CNN_PARAMS = ("fc0_w", "fc0_b", "fc2_w", "fc2_b", "d0_w", "d0_b", "dbn_g", "dbn_b", "d3_w", "d3_b")


class HeadsCnnParams(C.Structure):          # field order of tmf_heads_cnn_params (the two running buffers sit ahead of d3_*)
    _fields_ = [(n, _p) for n in HEADS_CNN_PARAMS[:8] + ("dbn_rm", "dbn_rv") + HEADS_CNN_PARAMS[8:]]


class HeadsCnnGrads(C.Structure):
    _fields_ = [(n, _p) for n in HEADS_CNN_PARAMS]


PROTOTYPES.update({
    "tmf_heads_cnn_saved_bytes": (_z, [C.POINTER(HeadsCnnDesc)]),
    "tmf_heads_cnn_bwd_scratch_bytes": (_z, [C.POINTER(HeadsCnnDesc)]),
    "tmf_heads_cnn_fwd": (_i, [C.POINTER(HeadsCnnDesc), _p, _p, C.POINTER(HeadsCnnParams), _p, _p, _p, _p, _z, _p]),
    "tmf_heads_cnn_bwd": (_i, [C.POINTER(HeadsCnnDesc), C.POINTER(HeadsCnnParams), _p, _z, _p, _p, _p,
                               C.POINTER(HeadsCnnGrads), _p, _p, _f, _p, _z, _p]),
})

_lib = None


class TmfError(RuntimeError):
    pass


def load():
    """dlopen libtmf_hip.so (once) and attach prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("TMF_LIB", LIB_PATH)      # TMF_LIB: another build of the same library (kernel A/B runs)
    if not os.path.exists(path):
        raise TmfError(
            f"{path} is missing: build it with `python -m transmf_ad_amd.build` "
            "(hipcc --offload-arch=gfx950). transmf_ad_amd has no CPU or PyTorch fallback.")
    if "TMF_LIB" not in os.environ and os.environ.get("TMF_SKIP_STAMP_CHECK", "0") != "1":
        # a library built from other sources / other compile flags than the ones next to it must not run silently
        from . import build as _build
        try:
            with open(path + ".stamp") as f:
                stamp = f.read().strip()
        except OSError:
            stamp = None
        if stamp != _build.source_digest():
            raise TmfError(
                f"{path} is stale: it was not built from the current csrc/*.hip + headers + compile flags "
                "(stamp mismatch). Rebuild with `python -m transmf_ad_amd.build`.")
    lib = C.CDLL(path)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def pool_code(pool):
    return _POOL[pool]


def call(name, *args):
    """Call an int-returning entry point; raise TmfError with the library's message on failure."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.tmf_last_error_string()
        raise TmfError(f"{name} failed (rc={rc}): {msg.decode() if msg else ''}")


def query(name, *args):
    """Call a size/count query (no error code)."""
    return getattr(load(), name)(*args)
