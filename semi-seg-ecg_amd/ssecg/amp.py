"""Reduced-precision training path (``use_amp: true``).

Reference semantics: the STUDENT forward of every plugin runs under ``torch.cuda.amp.autocast``
(``src/algorithms/fixmatch.py:97``, ``mean_teacher.py:98``, ``base.py:122``, ``cps.py:115``, ``stpp.py:159``):
convolutions in 16 bit with fp32 accumulation, BatchNorm statistics and the loss in fp32, master weights fp32;
the teacher / pseudo-label passes are outside autocast and stay fp32.

Here (``enable(model)``): train-mode forwards of the ResNet body and the FCN head's conv unit keep every
activation in HBM as **bf16 in the blocked layout** ``(N, C/8, L, 8)`` and run their convolutions on
``v_mfma_f32_32x32x16_bf16`` (``csrc/amp.hip``); roundings sit where PyTorch's autocast puts them (conv output, BN output,
the sum ``out += identity``; each branch's input gradient before autograd adds them), gradients are rounded where
they are stored.  The C-lead stem runs on 16-bit operands as under autocast (round 5: fp32 MFMA on bf16-rounded x and w, rounded
output).  What stays fp32 (documented deviations, all towards more precision): dropout + the 1x1 classifier + interpolation +
losses, all statistics, all weight gradients, AdamW / EMA.  The pseudo-label / teacher passes of the training steps are outside
autocast in the reference and fp32 here; ``evaluate()`` / ``test()`` / ``inference.py`` run INSIDE it (``base.py:202``) and take the
16-bit eval path below (``eval_autocast``, round 6).  Parity: the reference's CUDA autocast cannot run without CUDA, PyTorch's
CPU bf16 autocast can - ``tests/golden/ampfix_*`` hold the reference's real ``train_one_epoch(use_amp=True)`` executed under
it; ``tests/test_ampfix_gpu.py`` holds these kernels to those vectors block by block (outputs up to isolated 1-ulp flips),
``oracle/amp_ref.py`` emulates the rounding points on the CPU and is pinned to the same vectors.
"""
from __future__ import annotations

import weakref

import torch
import torch.distributed as dist

from . import config
from . import functional as SF
from . import ops
from .lib import SsecgError, check, lib, trace

config.passthrough("SSECG_AMP_WS", "bf16 weights-stationary kernel: unset = where it measured faster, 0 = never, 1 = wherever it applies "
                                   "(read by csrc/amp_ws.hip per call)")
from .ops import _p, _stream, _Timed


def enable(model: torch.nn.Module, on: bool = True) -> torch.nn.Module:
    """Switch the train-mode forward of every hot-path module of ``model`` to the bf16 path."""
    for m in model.modules():
        if hasattr(m, "_ssecg_amp_capable"):
            m.amp = bool(on)
    return model


def is_blocked(t) -> bool:
    return isinstance(t, torch.Tensor) and t.dtype == torch.bfloat16 and t.dim() == 4 and t.shape[3] == 8


def _reqb(t, name):
    if not is_blocked(t) or not t.is_cuda:
        raise SsecgError(f"{name}: expected a blocked bf16 HIP tensor (N, C/8, L, 8)")
    return t if t.is_contiguous() else t.contiguous()


def to_blocked(x: torch.Tensor) -> torch.Tensor:
    """(N, C, L) fp32 -> (N, C/8, L, 8) bf16 (round to nearest even)."""
    x = ops._req(x, "x")
    N, C, L = x.shape
    if C % 8:
        raise SsecgError("to_blocked: channel count must be a multiple of 8")
    y = torch.empty((N, C // 8, L, 8), device=x.device, dtype=torch.bfloat16)
    with _Timed("cvt_planar_to_blocked_kernel", 0.0, 6.0 * x.numel()):
        check(lib().ssecg_amp_planar_to_blocked(_p(x), _p(y), N, C, L, _stream()), "ssecg_amp_planar_to_blocked")
    return y


def to_planar(xb: torch.Tensor) -> torch.Tensor:
    xb = _reqb(xb, "x")
    N, CB, L, _ = xb.shape
    y = torch.empty((N, CB * 8, L), device=xb.device, dtype=torch.float32)
    with _Timed("cvt_blocked_to_planar_kernel", 0.0, 6.0 * y.numel()):
        check(lib().ssecg_amp_blocked_to_planar(_p(xb), _p(y), N, CB * 8, L, _stream()), "ssecg_amp_blocked_to_planar")
    return y


# ----------------------------------------------------------------------------- weight operands
class _OpEntry:
    __slots__ = ("ref", "tag", "geom", "ops")

    def __init__(self, w, k, stride):
        Cout, Cin, K = w.shape
        self.ref = weakref.ref(w)
        self.tag = None
        self.geom = (Cout, Cin, K, stride)
        dev = w.device

        def buf(rows, ck, ntaps):
            return torch.empty(((ck // 16) * ntaps * 2 * rows, 8), device=dev, dtype=torch.bfloat16)

        # name -> (tensor, transposed, tap list)
        self.ops = {"fwd": (buf(Cout, Cin, K), 0, list(range(K)))}
        if stride == 1:
            self.ops["dg"] = (buf(Cin, Cout, K), 1, list(range(K)))
        elif K == 3:
            self.ops["dgA"] = (buf(Cin, Cout, 1), 1, [1])       # even input positions see tap 1
            self.ops["dgB"] = (buf(Cin, Cout, 2), 1, [0, 2])    # odd input positions see taps 0 and 2
        else:
            self.ops["dg"] = (buf(Cin, Cout, 1), 1, [0])


_cache = {}
_table = {}


def _refresh_all(device):
    ops._overlap_fence(device)      # (ops.PassOverlap: a lazy re-make inside an overlapped region is ordered on both streams)
    try:
        _refresh_all_unfenced(device)
    finally:
        ops._overlap_fence(device)


def _refresh_all_unfenced(device):
    live, rows, mx = [], [], 1
    for key, ent in list(_cache.items()):
        w = ent.ref()
        if w is None or w.data_ptr() != key or w.device != device:
            if w is None:
                del _cache[key]
            continue
        live.append(ent)
        Cout, Cin, K, _ = ent.geom
        for t, tr, taps in ent.ops.values():
            code = sum(tp << (8 * i) for i, tp in enumerate(taps))
            rows += [w.data_ptr(), t.data_ptr(), Cout, Cin, K, tr, len(taps), code]
            mx = max(mx, t.shape[0])
            ops.keep_for_graph(w, t)
    if not rows:       # every registered weight has been freed
        return
    key = tuple(rows)
    tab = ops.table_for(_table, 0, key, rows, device)
    check(lib().ssecg_amp_weight_operand_multi(_p(tab), len(rows) // 8, mx, _stream()), "ssecg_amp_weight_operand_multi")
    for ent in live:
        ent.tag = ops._weights_epoch[0]


def operand(w, name, stride):
    """bf16 MFMA operand ``name`` of weight ``w``; all registered weights are re-made in ONE launch the first time one is
    asked for after ``ops.begin_forward()`` (same trust rule as the Winograd operands, ssecg/ops.py)."""
    key = w.data_ptr()
    ent = _cache.get(key)
    if ent is None or ent.ref() is not w or ent.geom != (w.shape[0], w.shape[1], w.shape[2], stride):
        if len(_cache) > 512:
            _cache.clear()
        ent = _cache[key] = _OpEntry(w, w.shape[2], stride)
    if ent.tag != ops._weights_epoch[0]:
        _refresh_all(w.device)
        if ent.tag != ops._weights_epoch[0]:
            raise SsecgError("internal: bf16 operand cache did not refresh")
    return ent.ops[name][0]


# ----------------------------------------------------------------------------- kernels
def conv_fwd(xb, w, stride, pad, want_stats=True):
    """nn.Conv1d(k in {1,3}, stride, pad, dilation 1, no bias) on blocked bf16 -> (y blocked bf16, stats partial rows)."""
    xb = _reqb(xb, "x")
    N, CBi, Lin, _ = xb.shape
    Cout, Cin, K = w.shape
    if Cin != CBi * 8 or K not in (1, 3) or Cin % 16 or Cout % 64:
        raise SsecgError("amp.conv_fwd: unsupported shape")
    Lout = ops.conv_out_len(Lin, K, stride, pad, 1)
    y = torch.empty((N, Cout // 8, Lout, 8), device=xb.device, dtype=torch.bfloat16)
    Lb = lib()
    parts, stats = 0, None
    tap = [t - pad for t in range(K)] + [0, 0]
    if want_stats:
        parts = Lb.ssecg_amp_conv_parts(N, Cin, Lin, Cout, Lout, K, stride, tap[0], tap[1], tap[2], Lout, 1, 0)
        stats = torch.empty((parts, Cout, 2), device=xb.device, dtype=torch.float32)
    trace("amp.conv_fwd", tuple(xb.shape), tuple(w.shape), stride, pad)
    with _Timed(f"conv_b16_kernel<{K}> fwd", 2.0 * N * Lout * Cout * Cin * K, 2.0 * (xb.numel() + y.numel() + w.numel())):
        check(Lb.ssecg_amp_conv(_p(xb), _p(operand(w, "fwd", stride)), _p(y), N, Cin, Lin, Cout, Lout, K, stride, tap[0], tap[1],
                                tap[2], Lout, 1, 0, None, _p(stats), parts, _stream()), "ssecg_amp_conv")
    return y, stats


def conv_dgrad(dyb, w, in_len, stride, pad, accumulate=None, inplace=False):
    """dx (blocked bf16) of conv_fwd; ``accumulate`` (blocked like dx) is added to the ROUNDED result and the sum rounded again (autograd's bf16 add of two stored gradients); ``inplace`` (1x1 stride-2
    only): the sum is written back into ``accumulate`` at the even positions, the odd ones keep what it holds."""
    dyb = _reqb(dyb, "dy")
    N, CBo, Lout, _ = dyb.shape
    Cout, Cin, K = w.shape
    if Cout != CBo * 8 or Cout % 16 or Cin % 64:
        raise SsecgError("amp.conv_dgrad: unsupported shape")
    Lb = lib()
    if accumulate is not None:
        accumulate = _reqb(accumulate, "accumulate")
        if tuple(accumulate.shape) != (N, Cin // 8, in_len, 8):
            raise SsecgError("amp.conv_dgrad: accumulate shape mismatch")
    flops = 2.0 * N * Lout * Cout * Cin * K
    nbytes = 2.0 * (dyb.numel() + N * Cin * in_len * (2 if accumulate is not None else 1))
    trace("amp.conv_dgrad", tuple(dyb.shape), tuple(w.shape), in_len, stride, pad)
    if stride == 1:
        dx = torch.empty((N, Cin // 8, in_len, 8), device=dyb.device, dtype=torch.bfloat16)
        tap = [pad - t for t in range(K)] + [0, 0]
        with _Timed(f"conv_b16_kernel<{K}> dgrad", flops, nbytes):
            check(Lb.ssecg_amp_conv(_p(dyb), _p(operand(w, "dg", 1)), _p(dx), N, Cout, Lout, Cin, in_len, K, 1, tap[0], tap[1], tap[2],
                                    in_len, 1, 0, _p(accumulate), None, 0, _stream()), "ssecg_amp_conv")
        return dx
    if stride != 2 or not ((K == 3 and pad == 1) or (K == 1 and pad == 0)):
        raise SsecgError("amp.conv_dgrad: unsupported stride / padding")
    n_even, n_odd = (in_len + 1) // 2, in_len // 2
    if K == 1:
        # only even input positions receive a gradient; the odd ones are zero (or the accumulated tensor)
        if accumulate is None:
            dx = torch.zeros((N, Cin // 8, in_len, 8), device=dyb.device, dtype=torch.bfloat16)
        else:
            dx = accumulate if inplace else accumulate.clone()
        with _Timed("conv_b16_kernel<1> dgrad (stride 2)", flops, nbytes):
            check(Lb.ssecg_amp_conv(_p(dyb), _p(operand(w, "dg", 2)), _p(dx), N, Cout, Lout, Cin, n_even, 1, 1, 0, 0, 0, in_len, 2, 0,
                                    _p(dx if accumulate is not None else None), None, 0, _stream()), "ssecg_amp_conv")
        return dx
    dx = torch.empty((N, Cin // 8, in_len, 8), device=dyb.device, dtype=torch.bfloat16)
    with _Timed("conv_b16_kernel<1> + <2> dgrad (stride-2 phases)", flops, nbytes):
        # even inputs l = 2j: tap 1 from dy[j];  odd inputs l = 2j+1: tap 0 from dy[j+1], tap 2 from dy[j]
        check(Lb.ssecg_amp_conv(_p(dyb), _p(operand(w, "dgA", 2)), _p(dx), N, Cout, Lout, Cin, n_even, 1, 1, 0, 0, 0, in_len, 2, 0,
                                _p(accumulate), None, 0, _stream()), "ssecg_amp_conv")
        if n_odd > 0:
            check(Lb.ssecg_amp_conv(_p(dyb), _p(operand(w, "dgB", 2)), _p(dx), N, Cout, Lout, Cin, n_odd, 2, 1, 1, 0, 0, in_len, 2, 1,
                                    _p(accumulate), None, 0, _stream()), "ssecg_amp_conv")
    return dx


def conv_wgrad(dyb, xb, ksize, stride, pad):
    """dw (Cout, Cin, K) fp32 from blocked bf16 operands."""
    dyb = _reqb(dyb, "dy"); xb = _reqb(xb, "x")
    N, CBo, Ldy, _ = dyb.shape
    _, CBi, Lx, _ = xb.shape
    Cout, Cin = CBo * 8, CBi * 8
    Lb = lib()
    if Lb.ssecg_amp_wgrad_supported(N, Cin, Lx, Cout, Ldy, ksize, stride, pad) != 1:
        raise SsecgError("amp.conv_wgrad: unsupported shape")
    nbytes = Lb.ssecg_amp_wgrad_workspace(N, Cin, Lx, Cout, Ldy, ksize)
    ws = ops._workspace(xb.device, nbytes)
    dw = torch.empty((Cout, Cin, ksize), device=xb.device, dtype=torch.float32)
    trace("amp.conv_wgrad", tuple(dyb.shape), tuple(xb.shape), ksize, stride, pad)
    with _Timed("conv_wgrad_b16_kernel + wgrad_b16_reduce_kernel", 2.0 * N * Ldy * Cout * Cin * ksize,
                2.0 * (dyb.numel() + xb.numel()) + 4.0 * dw.numel()):
        check(Lb.ssecg_amp_wgrad(_p(dyb), _p(xb), _p(dw), N, Cin, Lx, Cout, Ldy, ksize, stride, pad, _p(ws), ws.numel(), _stream()),
              "ssecg_amp_wgrad")
    return dw


def bn_apply_fwd(xb, mean, invstd, gamma, beta, residual=None, relu=True, want_mask=False, res_bn=None):
    """-> y, or (y, mask) with ``want_mask``: uint8 (N, C/8, L), one byte per 16-byte vector of y, bit j = (channel 8*cb + j > 0).
    ``res_bn`` = (mean, invstd, gamma, beta): ``residual`` is the raw output of the 1x1 downsample convolution; its BatchNorm is applied
    (and rounded to bf16, as the stored identity tensor was) while it is read."""
    xb = _reqb(xb, "x")
    N, CB, L, _ = xb.shape
    if residual is not None:
        residual = _reqb(residual, "residual")
        if tuple(residual.shape) != tuple(xb.shape):
            raise SsecgError("amp.bn_apply_fwd: residual shape mismatch")
    rb = [None] * 4 if res_bn is None else list(res_bn)
    y = torch.empty_like(xb)
    mask = torch.empty((N, CB, L), device=xb.device, dtype=torch.uint8) if want_mask else None
    with _Timed("bn_apply_fwd_b16_kernel", 0.0, 2.0 * xb.numel() * (3 if residual is not None else 2)):
        check(lib().ssecg_amp_bn_apply_fwd_resbn(_p(xb), _p(y), N, CB * 8, L, _p(mean), _p(invstd), _p(gamma), _p(beta), _p(residual),
                                                 _p(rb[0]), _p(rb[1]), _p(rb[2]), _p(rb[3]), int(relu), _p(mask), _stream()),
              "ssecg_amp_bn_apply_fwd_resbn")
    return (y, mask) if want_mask else y


def _mask_of(yb, xb, mode):
    """mode 1 (saved output, blocked bf16) or 3 (the byte mask of bn_apply_fwd, uint8 (N, C/8, L))."""
    if mode == 3:
        if yb.dtype != torch.uint8 or not yb.is_cuda or not yb.is_contiguous() or tuple(yb.shape) != tuple(xb.shape[:3]):
            raise SsecgError("mask: expected the uint8 (N, C/8, L) mask of bn_apply_fwd on the HIP device")
    return yb


def bn_bwd_reduce(dyb, yb, xb, mean, invstd, gamma, beta, mode):
    dyb = _reqb(dyb, "dy"); xb = _reqb(xb, "x"); yb = _mask_of(yb, xb, mode)
    N, CB, L, _ = xb.shape
    Lb = lib()
    parts = Lb.ssecg_amp_bn_bwd_parts(N, CB * 8, L)
    partial = torch.empty((parts, CB * 8, 2), device=xb.device, dtype=torch.float32)
    with _Timed("bn_bwd_b16_kernel<reduce>", 0.0, 2.0 * xb.numel() * (3 if mode == 1 else 2)):
        check(Lb.ssecg_amp_bn_bwd_reduce(_p(dyb), _p(yb), _p(xb), _p(mean), _p(invstd), _p(gamma), _p(beta), mode, N, CB * 8, L,
                                         _p(partial), _stream()), "ssecg_amp_bn_bwd_reduce")
    return partial


def bn_bwd_apply(dyb, yb, xb, mean, invstd, gamma, beta, mode, sums, count, want_dz=False):
    dyb = _reqb(dyb, "dy"); xb = _reqb(xb, "x"); yb = _mask_of(yb, xb, mode)
    N, CB, L, _ = xb.shape
    dx = torch.empty_like(xb)
    dz = torch.empty_like(xb) if want_dz else None
    with _Timed("bn_bwd_b16_kernel<apply>", 0.0, 2.0 * xb.numel() * ((3 if mode == 1 else 2) + (2 if want_dz else 1))):
        check(lib().ssecg_amp_bn_bwd_apply(_p(dyb), _p(yb), _p(xb), _p(mean), _p(invstd), _p(gamma), _p(beta), mode, _p(sums),
                                           float(count), N, CB * 8, L, _p(dx), _p(dz), _stream()), "ssecg_amp_bn_bwd_apply")
    return dx, dz


def bn_bwd_pair(u2, ud, dyb):
    """``functional.bn_bwd_pair`` on blocked bf16: the block's bn2 (``u2``, mask in ``u2.y``) and the BatchNorm of its 1x1 branch (``ud``) from
    the block's output gradient -> (dc2, dcd, dgamma2, dbeta2, dgamma_d, dbeta_d): one reduction, one all-reduce, one apply pass."""
    dyb = _reqb(dyb, "dy")
    N, CB, L, _ = u2.c.shape
    C = CB * 8
    mode = 3 if u2.y.dtype == torch.uint8 else 1
    ym = _mask_of(u2.y, u2.c, mode)
    Lb = lib()
    parts = Lb.ssecg_amp_bn_bwd_parts(N, C, L)
    p2 = torch.empty((parts, C, 2), device=dyb.device, dtype=torch.float32)
    pd = torch.empty((parts, C, 2), device=dyb.device, dtype=torch.float32)
    with _Timed("bn_bwd_b16_kernel<reduce> (pair)", 0.0, 2.0 * u2.c.numel() * 3):
        check(Lb.ssecg_amp_bn_bwd_reduce_pair(_p(dyb), _p(ym), mode, _p(u2.c), _p(u2.mean), _p(u2.invstd), _p(ud.c), _p(ud.mean), _p(ud.invstd),
                                              N, C, L, _p(p2), _p(pd), _stream()), "ssecg_amp_bn_bwd_reduce_pair")
    if u2.group is not None:
        both = torch.empty((2 * C, 2), device=dyb.device, dtype=torch.float64)
        s2, dg2, db2 = ops.bn_reduce_partials(p2, want_param_grads=True, out=both[:C])
        sd, dgd, dbd = ops.bn_reduce_partials(pd, want_param_grads=True, out=both[C:])
        SF._allreduce_sums(both, u2.group)
    else:
        s2, dg2, db2 = ops.bn_reduce_partials(p2, want_param_grads=True)
        sd, dgd, dbd = ops.bn_reduce_partials(pd, want_param_grads=True)
    dc2, dcd = torch.empty_like(u2.c), torch.empty_like(ud.c)
    with _Timed("bn_bwd_b16_kernel<apply> (pair)", 0.0, 2.0 * u2.c.numel() * 5):
        check(Lb.ssecg_amp_bn_bwd_apply_pair(_p(dyb), _p(ym), mode, _p(u2.c), _p(u2.mean), _p(u2.invstd), _p(u2.gamma), _p(s2), _p(ud.c),
                                             _p(ud.mean), _p(ud.invstd), _p(ud.gamma), _p(sd), float(u2.count), N, C, L, _p(dc2), _p(dcd),
                                             _stream()), "ssecg_amp_bn_bwd_apply_pair")
    return dc2, dcd, dg2, db2, dgd, dbd


# ----------------------------------------------------------------------------- units (conv -> BN -> [+res] -> [ReLU])
class _U:
    __slots__ = ("x", "w", "c", "y", "mean", "invstd", "gamma", "beta", "relu", "stride", "pad", "count", "group")


def _unit_conv(xb, w, bn, stride, pad, sums_out=None, reduce_now=True):
    """Convolution with statistics and, under SyncBatchNorm, the reduced sums (all-reduced here unless the caller merges them)."""
    c, partial = conv_fwd(xb, w, stride, pad, want_stats=True)
    sums = None
    if bn.group is not None:
        sums = ops.bn_reduce_partials(partial, out=sums_out)
        if reduce_now:
            SF._allreduce_sums(sums, bn.group)
    return c, partial, sums


def unit_fwd_pair(xb, a, b):
    """Two independent units on the same input (a block's first convolution and its 1x1 downsample branch): their SyncBatchNorm
    sums travel in ONE all-reduce (``functional.unit_fwd_train_pair_begin``).  a, b = (w, bn, stride, pad, relu)."""
    (wa, bna, sa, pa, ra), (wb, bnb, sb, pb, rb) = a, b
    apply_b = not SF.RESBN_IN_PLACE
    if bna.group is None or bna.group is not bnb.group:
        return unit_fwd(xb, wa, bna, sa, pa, ra, None), unit_fwd(xb, wb, bnb, sb, pb, rb, None, apply=apply_b)
    Ca, Cb = wa.shape[0], wb.shape[0]
    both = torch.empty((Ca + Cb, 2), device=xb.device, dtype=torch.float64)
    ca = _unit_conv(xb, wa, bna, sa, pa, both[:Ca], False)
    cb = _unit_conv(xb, wb, bnb, sb, pb, both[Ca:], False)
    SF._allreduce_sums(both, bna.group)
    return unit_fwd(xb, wa, bna, sa, pa, ra, None, conv=ca), unit_fwd(xb, wb, bnb, sb, pb, rb, None, conv=cb, apply=apply_b)


def unit_fwd(xb, w, bn: SF.BNState, stride, pad, relu=True, residual=None, conv=None, apply=True, res_bn=None):
    """``apply=False`` / ``res_bn``: as ``functional.unit_fwd_train_finish`` (the 1x1 downsample branch's BatchNorm is applied by bn2's
    apply pass while it reads the raw 1x1 output)."""
    c, partial, sums = conv if conv is not None else _unit_conv(xb, w, bn, stride, pad)
    count = c.shape[0] * c.shape[2]
    if sums is not None:
        count *= dist.get_world_size(bn.group)
        mean, invstd = ops.bn_finalize(sums, count, bn.eps, bn.momentum, bn.running_mean, bn.running_var)
    else:
        mean, invstd = ops.bn_stats_finalize(partial, count, bn.eps, bn.momentum, bn.running_mean, bn.running_var)
    SF._count_batch(bn.num_batches_tracked)
    # a unit with a residual cannot recompute its ReLU mask from c: the apply pass leaves a byte per vector for the backward
    # (1/16 of the bytes of the saved output); non-residual units recompute the mask from c
    if not apply:
        if relu or residual is not None:
            raise SsecgError("amp.unit_fwd: apply=False is for the 1x1 downsample branch")
        y, mask = None, None
    elif relu and residual is not None and ops.BN_MASK_BITS:
        y, mask = bn_apply_fwd(c, mean, invstd, bn.weight, bn.bias, residual, relu, want_mask=True, res_bn=res_bn)
    else:
        y, mask = bn_apply_fwd(c, mean, invstd, bn.weight, bn.bias, residual, relu, res_bn=res_bn), None
    u = _U()
    u.x, u.w, u.c = xb, w, c
    u.y = (mask if mask is not None else y) if (relu and residual is not None) else None
    u.mean, u.invstd, u.gamma, u.beta = mean, invstd, bn.weight, bn.bias
    u.relu, u.stride, u.pad, u.count, u.group = relu, stride, pad, count, bn.group
    return y, u


_NT = 8


def _pack(us):
    flat, metas = [], []
    for u in us:
        if u is None:
            metas.append(None)
            continue
        flat += [u.x, u.w, u.c, u.y, u.mean, u.invstd, u.gamma, u.beta]
        metas.append((u.relu, u.stride, u.pad, u.count, u.group))
    return flat, metas


def _unpack(saved, metas):
    out, k = [], 0
    for m in metas:
        if m is None:
            out.append(None)
            continue
        u = _U()
        u.x, u.w, u.c, u.y, u.mean, u.invstd, u.gamma, u.beta = saved[k:k + _NT]
        u.relu, u.stride, u.pad, u.count, u.group = m
        out.append(u)
        k += _NT
    return out, saved[k:]


def unit_bwd(u: _U, dyb, need_dx=True, dx_accumulate=None, need_dz=False, fill=None, defer_wgrad=False, dx_inplace=False, dy_mask=None):
    """-> (dx, dw, dgamma, dbeta, dz); ``fill`` / ``defer_wgrad`` / ``dy_mask`` as in ``functional.unit_bwd`` (the previous unit's
    weight gradient is launched inside this unit's SyncBN all-reduce window; a unit without a ReLU of its own takes its incoming
    gradient masked by another unit's ReLU mask while reading it)."""
    mode = 0 if not u.relu else (2 if u.y is None else (3 if u.y.dtype == torch.uint8 else 1))
    ymask = u.y
    if dy_mask is not None:
        if u.relu:
            raise SsecgError("amp.unit_bwd: dy_mask is for units without a ReLU of their own")
        ymask, mode = dy_mask, (3 if dy_mask.dtype == torch.uint8 else 1)
    partial = bn_bwd_reduce(dyb, ymask, u.c, u.mean, u.invstd, u.gamma, u.beta, mode)
    sums, dgamma, dbeta = ops.bn_reduce_partials(partial, want_param_grads=True)
    if u.group is not None and fill is not None:
        sums, work = SF._allreduce_sums_async(sums, u.group)
        fill()
        work.wait()
    elif u.group is not None:    # nothing to overlap: the synchronous form runs on this stream (no cross-stream hand-offs)
        SF._allreduce_sums(sums, u.group)
    elif fill is not None:
        fill()
    dc, dz = bn_bwd_apply(dyb, ymask, u.c, u.mean, u.invstd, u.gamma, u.beta, mode, sums, u.count, want_dz=need_dz)

    def launch_wgrad():
        return conv_wgrad(dc, u.x, u.w.shape[2], u.stride, u.pad)

    dw = launch_wgrad if defer_wgrad else launch_wgrad()
    dx = conv_dgrad(dc, u.w, u.x.shape[2], u.stride, u.pad, accumulate=dx_accumulate, inplace=dx_inplace) if need_dx else None
    return dx, dw, dgamma, dbeta, dz


class BasicBlockAmpFn(torch.autograd.Function):
    """BasicBlock.forward (src/models/backbones/resnet.py:55-72) on blocked bf16 activations, train mode."""

    @staticmethod
    def forward(ctx, x, w1, g1, b1, w2, g2, b2, wd, gd, bd, bn1, bn2, bnd, stride):
        if wd is not None:
            (a1, u1), (idt, ud) = unit_fwd_pair(x, (w1, bn1, stride, 1, True), (wd, bnd, stride, 0, False))
        else:
            (a1, u1), (idt, ud) = unit_fwd(x, w1, bn1, stride, 1, True, None), (x, None)
        res_bn = None
        if wd is not None and idt is None:    # (SF.RESBN_IN_PLACE) bn2's apply pass normalises the raw 1x1 output while reading it
            idt, res_bn = ud.c, (ud.mean, ud.invstd, ud.gamma, ud.beta)
        out, u2 = unit_fwd(a1, w2, bn2, 1, 1, True, idt, res_bn=res_bn)
        flat, ctx.metas = _pack([u1, u2, ud])
        ctx.save_for_backward(*flat)
        ctx.has_ds = wd is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        (u1, u2, ud), _ = _unpack(ctx.saved_tensors, ctx.metas)
        dout = dout.contiguous()
        got = {}
        dz_in_place = SF.DZ_IN_PLACE and ctx.has_ds and not SF.DS_BRANCH_FIRST and u2.y is not None    # (functional.BasicBlockFn.backward)
        pending = []

        def run_pending():
            while pending:
                name, fn = pending.pop(0)
                got[name] = fn()

        if (dz_in_place and SF.PAIR_DS_BWD and u2.group is ud.group and u2.count == ud.count and tuple(u2.c.shape) == tuple(ud.c.shape)):
            # bn2 and the 1x1 branch's BatchNorm see the same masked gradient: one reduction, one all-reduce, one apply pass for both
            dc2, dcd, dg2, db2, dgd, dbd = bn_bwd_pair(u2, ud, dout)
            pending.append(("w2", lambda: conv_wgrad(dc2, u2.x, u2.w.shape[2], u2.stride, u2.pad)))
            da1 = conv_dgrad(dc2, u2.w, u2.x.shape[2], u2.stride, u2.pad)
            dx1, w1f, dg1, db1, _ = unit_bwd(u1, da1, need_dx=True, fill=run_pending, defer_wgrad=True)
            pending.append(("w1", w1f))
            pending.append(("wd", lambda: conv_wgrad(dcd, ud.x, ud.w.shape[2], ud.stride, ud.pad)))
            dx = conv_dgrad(dcd, ud.w, ud.x.shape[2], ud.stride, ud.pad, accumulate=dx1, inplace=True)
            run_pending()
            return dx, got["w1"], dg1, db1, got["w2"], dg2, db2, got["wd"], dgd, dbd, None, None, None, None
        da1, w2, dg2, db2, dz = unit_bwd(u2, dout, need_dx=True, need_dz=not dz_in_place, defer_wgrad=True)
        pending.append(("w2", w2))

        dwd = dgd = dbd = None
        if ctx.has_ds and SF.DS_BRANCH_FIRST:   # the order of rounds 2-3 (SSECG_DS_FIRST=1)
            acc, wd, dgd, dbd, _ = unit_bwd(ud, dz, need_dx=True, fill=run_pending, defer_wgrad=True)
            pending.append(("wd", wd))
            dx, dw1, dg1, db1, _ = unit_bwd(u1, da1, need_dx=True, dx_accumulate=acc, fill=run_pending)
            run_pending()
            dwd = got["wd"]
        elif ctx.has_ds:
            # main branch first (functional.BasicBlockFn.backward): its two stride-2 phases write every position, rounded once; the
            # 1x1 branch's gradient is then added in place at the even positions (the sum rounded again) - no zero fill of dx, no
            # accumulate reads in the phases, and the even phase is a plain 1-tap launch the weights-stationary kernel takes
            dx1, w1f, dg1, db1, _ = unit_bwd(u1, da1, need_dx=True, fill=run_pending, defer_wgrad=True)
            pending.append(("w1", w1f))
            dx, wd, dgd, dbd, _ = unit_bwd(ud, dout if dz_in_place else dz, need_dx=True, dx_accumulate=dx1, dx_inplace=True, fill=run_pending,
                                           defer_wgrad=True, dy_mask=u2.y if dz_in_place else None)
            pending.append(("wd", wd))
            run_pending()
            dw1, dwd = got["w1"], got["wd"]
        else:
            dx, dw1, dg1, db1, _ = unit_bwd(u1, da1, need_dx=True, dx_accumulate=dz, fill=run_pending)
            run_pending()
        dw2 = got["w2"]
        return dx, dw1, dg1, db1, dw2, dg2, db2, dwd, dgd, dbd, None, None, None, None


class UnitAmpFn(torch.autograd.Function):
    """One conv -> BN -> ReLU unit on blocked bf16 (the FCN head's conv unit, fcn_head.py:39-47)."""

    @staticmethod
    def forward(ctx, xb, w, gamma, beta, bn, stride, pad, relu):
        y, u = unit_fwd(xb, w, bn, stride, pad, relu, None)
        flat, ctx.metas = _pack([u])
        ctx.save_for_backward(*flat)
        return y

    @staticmethod
    def backward(ctx, dyb):
        (u,), _ = _unpack(ctx.saved_tensors, ctx.metas)
        dx, dw, dg, db, _ = unit_bwd(u, dyb.contiguous(), need_dx=ctx.needs_input_grad[0])
        return dx, dw, dg, db, None, None, None, None


class ToBlockedFn(torch.autograd.Function):
    """fp32 (N, C, L) -> blocked bf16 (one rounding); the gradient comes back as fp32 planar (exact conversion)."""

    @staticmethod
    def forward(ctx, x):
        return to_blocked(x)

    @staticmethod
    def backward(ctx, dyb):
        return to_planar(dyb.contiguous())


class ToPlanarFn(torch.autograd.Function):
    """blocked bf16 -> fp32 (N, C, L) (exact); the gradient is rounded once into the blocked layout."""

    @staticmethod
    def forward(ctx, xb):
        return to_planar(xb)

    @staticmethod
    def backward(ctx, dy):
        return to_blocked(dy.contiguous())


# ----------------------------------------------------------------------------- eval mode inside autocast
# ``evaluate()`` / ``test()`` run their eval-mode forward INSIDE ``torch.cuda.amp.autocast(enabled=use_amp)``
# (src/algorithms/base.py:202, called with the config's ``use_amp`` at base.py:369-375, 476-481, fixmatch.py:338-344) - unlike the
# pseudo-label passes of the training steps, which are outside it (fixmatch.py:87-91, mean_teacher.py:90-92, cps.py:95-101).
# ``eval_autocast(model, use_amp)`` selects, for the duration of one forward, the 16-bit eval path: the train path's kernels with the
# running statistics (conv output rounded, BatchNorm output rounded, ``out += identity`` rounded again - autocast's placements);
# dropout is the identity, the 1x1 classifier / interpolation / loss stay fp32 as in the train path (documented deviation).
class eval_autocast:
    def __init__(self, model, on=True):
        self.mods = [m for m in model.modules() if hasattr(m, "_ssecg_amp_capable")] if on else []

    def __enter__(self):
        for m in self.mods:
            m.amp_eval = True
        return self

    def __exit__(self, *exc):
        for m in self.mods:
            m.amp_eval = False
        return False


def unit_fwd_eval(xb, w, bn: SF.BNState, stride, pad, relu=True, residual=None):
    """conv -> eval-mode BN -> [+residual] -> [ReLU] on blocked bf16 (no statistics, nothing saved)."""
    scale, shift = ops.bn_fold_cached(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)
    c, _ = conv_fwd(xb, w, stride, pad, want_stats=False)
    return bn_apply_fwd(c, None, None, scale, shift, residual, relu)


def block_forward_eval(block, xb):
    if block.dilation != 1:
        raise SsecgError("amp path: dilated BasicBlocks are not built (no shipped config uses them)")
    ds = block.downsample
    a1 = unit_fwd_eval(xb, block.conv1.weight, SF.BNState.of(block.bn1), block.stride, 1, True)
    idt = unit_fwd_eval(xb, ds[0].weight, SF.BNState.of(ds[1]), block.stride, 0, False) if ds is not None else xb
    return unit_fwd_eval(a1, block.conv2.weight, SF.BNState.of(block.bn2), 1, 1, True, idt)


def block_forward(block, xb):
    """BasicBlock on blocked bf16: train mode (the student pass) or, under ``eval_autocast``, eval mode."""
    if not block.training:
        if torch.is_grad_enabled() and xb.requires_grad:
            raise SsecgError("amp eval path: no backward through an eval-mode block")
        return block_forward_eval(block, xb)
    ds = block.downsample
    wd = gd = bd = bnd = None
    if ds is not None:
        wd, gd, bd, bnd = ds[0].weight, ds[1].weight, ds[1].bias, SF.BNState.of(ds[1])
    if block.dilation != 1:
        raise SsecgError("amp path: dilated BasicBlocks are not built (no shipped config uses them)")
    return BasicBlockAmpFn.apply(xb, block.conv1.weight, block.bn1.weight, block.bn1.bias, block.conv2.weight, block.bn2.weight,
                                 block.bn2.bias, wd, gd, bd, SF.BNState.of(block.bn1), SF.BNState.of(block.bn2), bnd, block.stride)
