"""transmf_ad_amd — MI355X-native (gfx950) forward/backward hot path of TransMF_AD.

    from transmf_ad_amd import model_ad, model_CNN_ad, model_single

are drop-ins for the reference's ``models.mymodel`` classes (same constructors, forward
signatures, state_dict keys).  All hot-path compute runs in hand-written HIP kernels behind the
C ABI of ``include/tmf_hip.h`` (``libtmf_hip.so``, built by ``python -m transmf_ad_amd.build``);
there is no CPU / stock-PyTorch fallback.
"""
from ._lib import LIB_PATH, TmfError, load as load_library          # noqa: F401
from .gradient_reversal import GradientReversal, revgrad            # noqa: F401
from .mymodel import model_ad, model_CNN_ad, model_single           # noqa: F401
from .ops import get_conv_precision, set_activation_storage, set_conv_precision   # noqa: F401
from .pipeline import DevicePrefetcher, scale_intensity_flip, rotate_zoom         # noqa: F401
from .nifti import read_nifti, write_nifti, nifti_batches         # noqa: F401
from . import optim                                                  # noqa: F401
from .networks import (Attention, CrossTransformer_MOD_AVG, FeedForward, PreNorm,   # noqa: F401
                       Transformer, sNet)

__all__ = ["model_ad", "model_CNN_ad", "model_single", "sNet", "CrossTransformer_MOD_AVG", "Transformer",
           "Attention", "PreNorm", "FeedForward", "revgrad", "GradientReversal", "load_library", "TmfError",
           "set_conv_precision", "get_conv_precision", "set_activation_storage", "DevicePrefetcher", "scale_intensity_flip", "rotate_zoom",
           "read_nifti", "write_nifti", "nifti_batches"]
