"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/ssecg.h declares
(no compute calls here - there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "ssecg.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ssecg_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from ssecg.lib import LIB_PATH, SIGNATURES, lib
    if not os.path.exists(LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    syms = header_symbols()
    assert len(syms) >= 30
    assert sorted(SIGNATURES) == syms, "ctypes table and include/ssecg.h disagree"
    handle = lib()
    raw = ctypes.CDLL(LIB_PATH)
    for s in syms:
        assert getattr(raw, s) is not None
    assert handle.ssecg_abi_version() == 11
    assert handle.ssecg_build_arch() == b"gfx950"


def test_pure_host_queries():
    from ssecg.lib import lib
    L = lib()
    assert L.ssecg_conv1d_stats_parts(1024, 64, 64, 500, 3) > 0
    assert L.ssecg_conv1d_stats_parts(0, 64, 64, 500, 3) == -1
    assert L.ssecg_conv1d_wgrad_workspace(1024, 512, 63, 512, 63, 3) >= 512 * 512 * 3 * 4
    assert L.ssecg_bn_bwd_parts(8, 64, 500) >= 1 and L.ssecg_ce_parts(4, 2000) >= 1


def test_invalid_arguments_are_rejected_before_any_launch():
    from ssecg.lib import lib
    L = lib()
    # null pointers / bad shapes return SSECG_E_INVAL (-1) without touching a device
    assert L.ssecg_conv1d_fwd(None, None, None, 1, 1, 8, 1, 8, 3, 1, 1, 1, None, None, None, 0, None, 0, None, None, None, 0, None) == -1
    assert L.ssecg_conv1d_fwd(8, 8, 8, 1, 1, 8, 1, 8, 5, 1, 2, 1, None, None, None, 0, None, 0, None, None, None, 0, None) == -1  # k=5 unsupported
    assert L.ssecg_conv1d_fwd(8, 8, 8, 1, 1, 8, 1, 9, 3, 1, 1, 1, None, None, None, 0, None, 0, None, None, None, 0, None) == -1  # wrong Lout
    assert L.ssecg_maxpool1d_fwd(8, 8, 1, 10, 4, 3, 2, 1, None) == -1
    assert L.ssecg_adamw_multi(None, 1, 1, 1e-3, 0.9, 0.999, 1e-8, 0.05, 1, None, None, None, None) == -1
    assert L.ssecg_adamw_multi(8, 1, 1, 1e-3, 0.9, 0.999, 1e-8, 0.05, 0, None, None, None, None) == -1           # step counts from 1
    assert L.ssecg_sgd_multi(None, 1, 1, 1e-3, 0.9, 0.0, 1, None, None, None) == -1
    assert L.ssecg_grad_norm_workspace(65, 512 * 512 * 3) >= 65 * 2 * 4
    assert L.ssecg_grad_norm_multi(8, 2, 5, 1, 4, 100, 8, 4, 8, None, 2.0, 0.5, 2000, None) == -2              # workspace too small
    # maximum sizes: operands are addressed with 32-bit byte offsets -> tensors of 2 GiB or more are refused, not wrapped
    big_n = 2 ** 31 // (64 * 500 * 4) + 1                      # (N, 64, 500) fp32 just above 2 GiB
    assert L.ssecg_conv1d_fwd(8, 8, 8, big_n, 64, 500, 64, 500, 3, 1, 1, 1, None, None, None, 0, None, 0, None, None, None, 0, None) == -1
    assert L.ssecg_conv1d_wino_supported(big_n, 64, 500, 64) == 0 and L.ssecg_conv1d_wino_supported(1024, 64, 500, 64) == 1
    assert L.ssecg_conv1d_wino(8, 16, 8, big_n, 64, 500, 64, None, None, None, 0, None, 0, None, None, None, 0, None) == -1
    assert L.ssecg_conv1d_wino_wgrad_supported(big_n, 128, 250, 128) == 0
    assert L.ssecg_conv1d_wino_wgrad_workspace(big_n, 128, 250, 128) == 0
    # Winograd entry points: channel-count requirements, workspace contract
    assert L.ssecg_conv1d_wino_supported(4, 12, 100, 64) == 0 and L.ssecg_conv1d_wino_supported(4, 16, 100, 96) == 0
    assert L.ssecg_conv1d_wino_wgrad_supported(4, 64, 100, 128) == 0 and L.ssecg_conv1d_wino_wgrad_supported(4, 256, 100, 128) == 1
    need = L.ssecg_conv1d_wino_wgrad_workspace(4, 128, 100, 128)
    assert need > 0 and L.ssecg_conv1d_wino_wgrad(8, 8, 8, 4, 128, 100, 128, 8, need - 1, None, None, None) == -2      # SSECG_E_WORKSPACE
    assert L.ssecg_conv1d_wino(8, 16, 8, 4, 64, 100, 64, None, None, None, 0, None, 0, 8, None, None) == -1   # scale without shift
    # record pipeline / metrics
    assert L.ssecg_strong_augment(8, 8, 8, None, None, 2, 2, 4097, 0.5, 250.0, 1.0, 0.5, 0, None) == -1       # L > 4096
    assert L.ssecg_standardize(None, 8, 2, 10, None) == -1
    assert L.ssecg_seg_confusion(8, 8, 2, 33, 10, 8, None) == -1                                            # K > 32


def test_product_path_fails_loudly_on_cpu_tensors():
    import torch
    from ssecg import ops
    from ssecg.lib import SsecgError
    with pytest.raises(SsecgError):
        ops.conv1d_fwd(torch.zeros(1, 1, 8), torch.zeros(1, 1, 3), 1, 1)
    from helpers import build_hip_model
    from ssecg import synth
    model = build_hip_model(1, synth.model_state(0, 1), torch.device("cpu"))
    with pytest.raises(SsecgError):
        model(torch.zeros(1, 1, 2000))
