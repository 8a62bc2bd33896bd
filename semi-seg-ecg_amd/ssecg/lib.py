"""ctypes binding of ``include/ssecg.h`` (the C ABI of the gfx950 hot path).

The product path has NO CPU fallback: if ``libssecg_hip.so`` is missing (or a
symbol is absent) every entry point raises - build it with
``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C semi-seg-ecg_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SSECG_LIB overrides the path (A/B of kernel build variants); the default is the in-tree build
LIB_PATH = os.environ.get("SSECG_LIB") or os.path.join(_HERE, "libssecg_hip.so")

_vp, _i, _f, _d, _sz, _u64, _i64 = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_size_t, C.c_uint64, C.c_int64

# name -> (restype, argtypes); mirrors include/ssecg.h one to one
SIGNATURES = {
    "ssecg_abi_version": (_i, []),
    "ssecg_build_arch": (C.c_char_p, []),
    "ssecg_conv1d_stats_parts": (_i, [_i, _i, _i, _i, _i]),
    "ssecg_conv1d_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "ssecg_conv1d_fwd_split_workspace": (_sz, [_i, _i, _i, _i, _i, _i]),
    "ssecg_conv1d_dgrad_split_workspace": (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    "ssecg_conv1d_transpose_weight": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "ssecg_conv1d_dgrad": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "ssecg_conv1d_wgrad_workspace": (_sz, [_i, _i, _i, _i, _i, _i]),
    "ssecg_conv1d_wgrad": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp, _vp, _vp]),
    "ssecg_conv1d_wino_supported": (_i, [_i, _i, _i, _i]),
    "ssecg_conv1d_wino_parts": (_i, [_i, _i, _i]),
    "ssecg_conv1d_wino_weight": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "ssecg_conv1d_wino_weight_multi": (_i, [_vp, _i, _i, _vp]),
    "ssecg_conv1d_wino": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp]),
    "ssecg_conv1d_wino4_supported": (_i, [_i, _i, _i, _i]),
    "ssecg_conv1d_wino4_parts": (_i, [_i, _i, _i]),
    "ssecg_conv1d_wino4_weight_multi": (_i, [_vp, _i, _i, _vp]),
    "ssecg_conv1d_wino4_split": (_i, [_i, _i, _i, _i]),
    "ssecg_conv1d_wino4": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "ssecg_conv1d_wino_wgrad_supported": (_i, [_i, _i, _i, _i]),
    "ssecg_conv1d_wino_wgrad_workspace": (_sz, [_i, _i, _i, _i]),
    "ssecg_conv1d_wino_wgrad": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp, _vp, _vp]),
    "ssecg_conv1d_wino_wgrad4_supported": (_i, [_i, _i, _i, _i]),
    "ssecg_conv1d_wino_wgrad4_workspace": (_sz, [_i, _i, _i, _i]),
    "ssecg_conv1d_wino_wgrad4": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp, _vp, _vp]),
    "ssecg_bn_reduce_partials": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp]),
    "ssecg_bn_stats_finalize": (_i, [_vp, _i, _i, _d, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ssecg_bn_finalize": (_i, [_vp, _i, _d, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ssecg_bn_fold": (_i, [_vp, _vp, _vp, _vp, _i, _f, _vp, _vp, _vp]),
    "ssecg_bn_fold_multi": (_i, [_vp, _i, _i, _vp]),
    "ssecg_bn_mask_supported": (_i, [_i, _i, _i]),
    "ssecg_bn_apply_fwd": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "ssecg_bn_apply_fwd_resbn": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "ssecg_bn_bwd_parts": (_i, [_i, _i, _i]),
    "ssecg_bn_bwd_reduce": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "ssecg_bn_bwd_apply": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _d, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "ssecg_bn_bwd_pair_supported": (_i, [_i, _i, _i]),
    "ssecg_bn_bwd_reduce_pair": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "ssecg_bn_bwd_apply_pair": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _d, _i, _i, _i, _vp, _vp, _vp]),
    "ssecg_bn_param_grads": (_i, [_vp, _i, _vp, _vp, _vp]),
    "ssecg_channel_sum": (_i, [_vp, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "ssecg_maxpool1d_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ssecg_maxpool1d_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ssecg_stem_supported": (_i, [_i, _i, _i]),
    "ssecg_stem_c16_supported": (_i, [_i, _i, _i]),
    "ssecg_stem_parts": (_i, [_i, _i]),
    "ssecg_stem_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _i, _vp]),
    "ssecg_stem_fwd_eval_pool": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "ssecg_stem_wgrad_workspace": (_sz, [_i, _i, _i]),
    "ssecg_stem_wgrad": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "ssecg_stem_fwd2": (_i, [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _i, _i, _vp]),
    "ssecg_stem_wgrad2": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp, _sz, _i, _vp]),
    "ssecg_bn_relu_maxpool_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "ssecg_bn_relu_maxpool_bwd_reduce": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "ssecg_bn_relu_maxpool_bwd_apply": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _d, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "ssecg_interp_linear_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "ssecg_interp_linear_bwd": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "ssecg_dropout_fwd": (_i, [_vp, _vp, _vp, _sz, _f, _u64, _vp, _vp]),
    "ssecg_mask_scale": (_i, [_vp, _vp, _vp, _sz, _f, _vp]),
    "ssecg_softmax_conf_argmax": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "ssecg_ce_parts": (_i, [_i, _i]),
    "ssecg_ce_hard_fwd_bwd": (_i, [_vp, _vp, _vp, _f, _i, _i, _i, _f, _vp, _vp, _vp]),
    "ssecg_ce_soft_fwd_bwd": (_i, [_vp, _vp, _i, _i, _i, _f, _vp, _vp, _vp]),
    "ssecg_seg_confusion": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "ssecg_sum_partials": (_i, [_vp, _i, _i, _f, _vp, _vp]),
    "ssecg_loss_pair_finish": (_i, [_vp, _i, _vp, _i, _f, _f, _vp, _vp]),
    "ssecg_strong_augment": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _d, _d, _d, _d, _u64, _vp]),
    "ssecg_standardize": (_i, [_vp, _vp, _i, _i, _vp]),
    "ssecg_amp_planar_to_blocked": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "ssecg_amp_blocked_to_planar": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "ssecg_amp_stem_pool_supported": (_i, [_i, _i, _i]),
    "ssecg_amp_stem_pool_fwd": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "ssecg_amp_weight_operand_multi": (_i, [_vp, _i, _i, _vp]),
    "ssecg_amp_conv_parts": (_i, [_i] * 13),
    "ssecg_amp_conv": (_i, [_vp, _vp, _vp] + [_i] * 13 + [_vp, _vp, _i, _vp]),
    "ssecg_amp_bn_apply_fwd": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "ssecg_amp_bn_apply_fwd_resbn": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "ssecg_amp_bn_bwd_parts": (_i, [_i, _i, _i]),
    "ssecg_amp_bn_bwd_reduce": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "ssecg_amp_bn_bwd_apply": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _d, _i, _i, _i, _vp, _vp, _vp]),
    "ssecg_amp_bn_bwd_reduce_pair": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "ssecg_amp_bn_bwd_apply_pair": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _d, _i, _i, _i, _vp, _vp, _vp]),
    "ssecg_amp_wgrad_supported": (_i, [_i] * 8),
    "ssecg_amp_wgrad_workspace": (_sz, [_i] * 6),
    "ssecg_amp_wgrad": (_i, [_vp, _vp, _vp] + [_i] * 8 + [_vp, _sz, _vp]),
    "ssecg_adamw_coefficients": (_i, [_d, _d, _d, _d, _i, _vp]),
    "ssecg_adamw_multi": (_i, [_vp, _i, _i64, _d, _d, _d, _d, _d, _i, _vp, _vp, _vp, _vp]),
    "ssecg_sgd_multi": (_i, [_vp, _i, _i64, _d, _d, _d, _i, _vp, _vp, _vp]),
    "ssecg_grad_norm_workspace": (_sz, [_i, _i64]),
    "ssecg_grad_norm_multi": (_i, [_vp, _i, _i, _i, _i, _i64, _vp, _sz, _vp, _vp, _d, _d, _i, _vp]),
    "ssecg_grad_clip_multi": (_i, [_vp, _i, _i, _i, _i, _i64, _vp, _d, _vp]),
    "ssecg_ema_multi": (_i, [_vp, _i, _i64, _d, _vp]),
    "ssecg_pack_scaled_multi": (_i, [_vp, _i, _i64, _vp, _d, _vp]),
}

_lib = None


class SsecgError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load (once) and return the shared library with every prototype declared."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SsecgError(
                f"{LIB_PATH} not found: the MI355X hot path has no CPU fallback. Build it with "
                "`make -C semi-seg-ecg_amd/csrc` (hipcc --offload-arch=gfx950).")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        if handle.ssecg_abi_version() != 11:
            raise SsecgError("libssecg_hip.so ABI version mismatch")
        _lib = handle
    return _lib


_TRACE = os.environ.get("SSECG_TRACE") == "1"


def trace(what: str, *shape_info) -> None:
    """SSECG_TRACE=1: print every launch (name + shapes) to stderr before it is enqueued - with
    AMD_SERIALIZE_KERNEL=3 the last line before a GPU fault names the faulting kernel."""
    if _TRACE:
        import sys
        print("[ssecg]", what, *shape_info, file=sys.stderr, flush=True)


def check(code: int, what: str) -> None:
    if code == 0:
        return
    if code == -1:
        raise SsecgError(f"{what}: invalid argument (SSECG_E_INVAL)")
    if code == -2:
        raise SsecgError(f"{what}: workspace too small (SSECG_E_WORKSPACE)")
    raise SsecgError(f"{what}: HIP error {code}")
