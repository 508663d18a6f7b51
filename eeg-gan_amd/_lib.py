"""ctypes binding of ``libeeggan_hip.so`` (the C ABI declared in ``include/eeggan_hip.h``).

The library is the only compute back end of this package: if it cannot be loaded the
import of any compute entry point raises -- there is no CPU or eager-PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

# PyTorch ships its own copy of the HIP runtime (soname libamdhip64.so.7).  It must be in the
# process BEFORE this library is opened, so that the library's dependency on that soname binds to
# the same runtime instance that owns torch's streams and allocations; loaded in the other order
# the process ends up with two runtimes and every launch on a torch stream fails.
import torch  # noqa: F401  (side effect: loads torch/lib/libamdhip64.so)

_HERE = os.path.dirname(os.path.abspath(__file__))
# (EEGGAN_LIB: another build of the same library, for same-session comparisons of two builds)
LIB_PATH = os.environ.get("EEGGAN_LIB") or os.path.join(_HERE, "libeeggan_hip.so")

_f = C.c_void_p      # device float*
_i = C.c_int
_fl = C.c_float
_sz = C.c_size_t
_st = C.c_void_p     # hipStream_t

# name -> (restype, argtypes); every symbol include/eeggan_hip.h declares
SIGNATURES = {
    "eeggan_last_error": (C.c_char_p, []),
    "eeggan_abi_version": (_i, []),
    "eeggan_conv1d": (_i, [_f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _fl, _i, _fl, _f, _f, _f, _fl, _st]),
    "eeggan_conv1d_pixelnorm_supported": (_i, [_i, _i, _i, _i, _i]),
    "eeggan_conv1d_pixelnorm": (_i, [_f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _fl, _fl, _fl, _st]),
    "eeggan_conv1d_flip_weights": (_i, [_f, _f, _i, _i, _i, _st]),
    "eeggan_conv1d_flip_weights_multi": (_i, [C.c_void_p, _i, _st]),
    "eeggan_set_matrix_precision": (_i, [_i]),
    "eeggan_get_matrix_precision": (_i, []),
    "eeggan_conv1d_wgrad_workspace": (_sz, [_i, _i, _i, _i, _i]),
    "eeggan_conv1d_wgrad": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _i, _fl, _f, _f, _fl, _fl, _fl, _f, _sz, _st]),
    "eeggan_conv1d_wgrad_partial": (_i, [_f, _f, _i, _i, _i, _i, _i, _i, _fl, _f, _f, _fl, _f, _sz, C.c_void_p, C.c_void_p,
                                         _st]),
    "eeggan_wgrad_reduce_multi": (_i, [C.c_void_p, _i, _st]),
    "eeggan_conv1d_wgrad_partial_multi": (_i, [C.c_void_p, _i, C.c_void_p, C.c_void_p, _st]),
    "eeggan_dwconv1d": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _i, _fl, _i, _st]),
    "eeggan_dwconv1d_wgrad": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _i, _fl, _i, _st]),
    "eeggan_head_fwd": (_i, [_f, _f, _f, _f, _f, _i, _i, _i, _i, _fl, _i, _fl, _st]),
    "eeggan_head_wgrad": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _fl, _st]),
    "eeggan_head_dgrad": (_i, [_f, _f, _f, _i, _i, _i, _i, _fl, _st]),
    "eeggan_tail_fwd": (_i, [_f, _f, _f, _f, _f, _i, _i, _i, _i, _fl, _i, _fl, _st]),
    "eeggan_tail_fwd_workspace": (_sz, [_i, _i, _i, _i]),
    "eeggan_tail_fwd_ws": (_i, [_f, _f, _f, _f, _f, _i, _i, _i, _i, _fl, _i, _fl, C.c_void_p, _sz, _st]),
    "eeggan_tail_dgrad": (_i, [_f, _f, _f, _i, _i, _i, _i, _fl, _st]),
    "eeggan_tail_wgrad": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _fl, _st]),
    "eeggan_lrelu_drop_bwd": (_i, [_f, _f, _f, _f, _i, _i, _i, _fl, _st]),
    "eeggan_pixelnorm_drop_fwd": (_i, [_f, _f, _f, _f, _i, _i, _i, _fl, _st]),
    "eeggan_pixelnorm_drop_bwd": (_i, [_f, _f, _f, _f, _f, _i, _i, _i, _fl, _st]),
    "eeggan_scaled_tanh": (_i, [_f, _f, _sz, _fl, _st]),
    "eeggan_tanh_bwd": (_i, [_f, _f, _f, _sz, _fl, _st]),
    "eeggan_axpby": (_i, [_f, _fl, _f, _fl, _f, _sz, _st]),
    "eeggan_blend": (_i, [_f, _f, _f, _i, _f, _sz, _st]),
    "eeggan_channel_window": (_i, [_f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _st]),
    "eeggan_mix_batch": (_i, [_f, _f, _f, _f, _i, _sz, _st]),
    "eeggan_pow_base": (_i, [_f, _f, _sz, _fl, _st]),
    "eeggan_lognormal_fill": (_i, [_f, _sz, _fl, C.c_ulonglong, C.c_uint, C.c_void_p, _st]),
    "eeggan_resample": (_i, [_f, _f, _f, _f, _sz, _i, _i, _i, _st]),
    "eeggan_resample_quads": (_i, [_f, _f, _f, _f, _sz, _i, _i, _i, _st]),
    "eeggan_mbstd_workspace": (_sz, [_i, _i, _i, _i]),
    "eeggan_mbstd_fwd": (_i, [_f, _f, _f, _i, _i, _i, _i, _st]),
    "eeggan_mbstd_fwd_static": (_i, [_f, _f, _i, _i, _i, _i, _st]),
    "eeggan_mbstd_bwd": (_i, [_f, _f, _f, _i, _i, _i, _i, _st]),
    "eeggan_mbstd_bwd_bwd": (_i, [_f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _st]),
    "eeggan_mbstd_fwd_window": (_i, [_f, _f, _f, _i, _i, _i, _i, _i, _i, _st]),
    "eeggan_mbstd_bwd_window": (_i, [_f, _f, _f, _i, _i, _i, _i, _i, _i, _st]),
    "eeggan_mbstd_bwd_bwd_window": (_i, [_f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _st]),
    "eeggan_wgan_gp_loss_fwd": (_i, [_f, _f, _f, _f, _f, _i, _sz, _fl, _fl, _fl, _st]),
    "eeggan_wgan_gp_loss_bwd": (_i, [_f, _f, _f, _f, _f, _f, _f, _i, _sz, _fl, _fl, _fl, _st]),
    "eeggan_feature_match_fwd": (_i, [_f, _f, _f, _f, _i, _i, _st]),
    "eeggan_feature_match_bwd": (_i, [_f, _f, _f, _i, _i, _st]),
    "eeggan_channel_sum": (_i, [_f, _f, _i, _i, _i, _fl, _st]),
    "eeggan_channel_affine": (_i, [_f, _f, _f, _f, _i, _i, _i, _st]),
    "eeggan_product": (_i, [_f, _f, _f, _sz, _st]),
    "eeggan_position_sum": (_i, [_f, _f, _i, _i, _i, _fl, _st]),
    "eeggan_position_scale": (_i, [_f, _f, _f, _i, _i, _i, _st]),
    "eeggan_spectral_norm_fwd": (_i, [_f, _f, _f, _f, _f, _f, _f, _f, _i, _i, _fl, _i, _st]),
    "eeggan_spectral_norm_bwd": (_i, [_f, _f, _f, _f, _f, _f, _f, _i, _i, _st]),
    "eeggan_bn_lrelu_drop_fwd": (_i, [_f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _i, _i, _i, _fl, _fl, _fl, _i, _st]),
    "eeggan_bn_lrelu_drop_bwd": (_i, [_f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _f, _i, _i, _i, _fl, _i, _st]),
    "eeggan_bmm": (_i, [_f, _f, _f, _i, _i, _i, _i, _i, _i, _fl, _st]),
    "eeggan_softmax_dim1_fwd": (_i, [_f, _f, _i, _i, _i, _st]),
    "eeggan_softmax_dim1_bwd": (_i, [_f, _f, _f, _i, _i, _i, _st]),
    "eeggan_softmax_dim1_bwd_dy": (_i, [_f, _f, _f, _f, _i, _i, _i, _st]),
    "eeggan_maxpool_fwd": (_i, [_f, _f, C.c_void_p, _sz, _i, _i, _st]),
    "eeggan_maxpool_select": (_i, [_f, C.c_void_p, _f, _sz, _i, _i, _i, _st]),
    "eeggan_dot_workspace": (_sz, []),
    "eeggan_dot": (_i, [_f, _f, _f, _sz, C.c_void_p, _sz, _st]),
    "eeggan_mean_scale": (_i, [_f, _f, _i, _fl, _st]),
    "eeggan_mean_scale_bwd": (_i, [_f, _f, _i, _fl, _st]),
    "eeggan_swd_patches": (_i, [_f, _f, _f, _i, _i, _i, _i, _i, _i, _st]),
    "eeggan_swd_normalize_workspace": (_sz, []),
    "eeggan_swd_normalize": (_i, [_f, _sz, _i, _fl, C.c_void_p, _sz, _st]),
    "eeggan_swd_project_keys": (_i, [_f, _f, C.c_void_p, _sz, _i, _i, _st]),
    "eeggan_swd_sort_workspace": (_sz, [_sz, _i]),
    "eeggan_swd_sort": (_i, [C.c_void_p, C.c_void_p, _sz, _i, C.c_void_p, _sz, _st]),
    "eeggan_swd_distance_workspace": (_sz, []),
    "eeggan_swd_distance": (_i, [C.c_void_p, C.c_void_p, _sz, _f, C.c_void_p, _sz, _st]),
    "eeggan_cpc_encoder": (_i, [_f] * 13 + [_i, _i, _i, _st]),
    "eeggan_pnorm_pool": (_i, [_f, _f, _i, _i, _i, _st]),
    "eeggan_dense": (_i, [_f, _f, _f, _f, _f, _f, _sz, _i, _i, _sz, _sz, _i, _fl, _fl, _i, _st]),
    "eeggan_gru_forward": (_i, [_f, _f, _f, _f, _i, _i, _i, _st]),
    "eeggan_feature_stats_workspace": (_sz, [_sz, _i]),
    "eeggan_feature_stats": (_i, [_f, _sz, _i, _f, _f, C.c_void_p, _sz, _st]),
    "eeggan_adam_step": (_i, [C.c_void_p, _i, _fl, _fl, _fl, _fl, _fl, _st]),
    "eeggan_multi_copy": (_i, [C.c_void_p, _i, _st]),
    "eeggan_multi_axpby": (_i, [C.c_void_p, _i, _fl, _fl, _st]),
    "eeggan_score_loss": (_i, [_i, _f, _f, _i, _fl, _f, _f, _f, _st]),
    "eeggan_scale_by_scalar": (_i, [_f, _f, _f, _sz, _st]),
    "eeggan_grad_penalty_fwd": (_i, [_f, _f, _f, _i, _sz, _fl, _fl, _i, _st]),
    "eeggan_grad_penalty_bwd": (_i, [_f, _f, _f, _f, _i, _sz, _fl, _fl, _i, _st]),
    # typed forms (bfloat16 activation storage): the float32 signatures plus storage flags
    "eeggan_conv1d_t": (_i, [_f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _fl, _i, _fl, _f, _f, _f, _fl, _i, _i, _st]),
    "eeggan_conv1d_pixelnorm_t": (_i, [_f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _fl, _fl, _fl, _i, _st]),
    "eeggan_resample_t": (_i, [_f, _f, _f, _f, _sz, _i, _i, _i, _i, _i, _st]),
    "eeggan_resample_quads_t": (_i, [_f, _f, _f, _f, _sz, _i, _i, _i, _i, _i, _st]),
    "eeggan_resample_quads_affine_t": (_i, [_f, _i, _i, _i, _f, _f, _sz, _i, _i, _i, _i, _i, _st]),
    "eeggan_lrelu_drop_bwd_t": (_i, [_f, _f, _f, _f, _i, _i, _i, _fl, _i, _st]),
    "eeggan_pixelnorm_drop_fwd_t": (_i, [_f, _f, _f, _f, _i, _i, _i, _fl, _i, _st]),
    "eeggan_pixelnorm_drop_bwd_t": (_i, [_f, _f, _f, _f, _f, _i, _i, _i, _fl, _i, _st]),
    "eeggan_axpby_t": (_i, [_f, _fl, _f, _fl, _f, _sz, _i, _st]),
    "eeggan_blend_t": (_i, [_f, _f, _f, _i, _f, _sz, _i, _st]),
    "eeggan_cast": (_i, [_f, _f, _sz, _i, _i, _st]),
}

ACT_NONE, ACT_LRELU, ACT_TANH, ACT_MASK = 0, 1, 2, 3


class EegganError(RuntimeError):
    """An entry point of libeeggan_hip.so rejected its arguments or its launch failed."""


_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load the shared library (once) and bind every declared symbol; raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). This package has no fallback path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def library_id() -> str:
    """First 12 hex digits of the SHA-256 of the library file in use: ties a measurement to the build it ran."""
    import hashlib
    with open(LIB_PATH, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:12]


def call(name: str, *args):
    """Call an int-returning entry point and raise EegganError with the library's message on failure."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise EegganError(lib.eeggan_last_error().decode("utf-8", "replace"))
