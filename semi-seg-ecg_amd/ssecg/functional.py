"""Autograd-level building blocks of the hot path, composed from the HIP kernels.

The differentiable unit is ``conv -> BatchNorm -> [+residual] -> [ReLU]``
(``src/models/backbones/resnet.py:55-72``): in train mode the conv kernel's
epilogue already produced the batch statistics, in eval mode BatchNorm is folded
into the conv epilogue and the unit is ONE kernel.  Autograd sees one node per
stem / BasicBlock / head (not per op), and each node's backward chains the units
by hand so that the residual branch's gradient is added inside the dgrad
epilogue instead of by a separate pass.

SyncBatchNorm (``src/algorithms/fixmatch.py:290-291``): when a process group is
given, the fp64 per-channel sums are all-reduced (RCCL on ROCm) between
``bn_reduce_partials`` and ``bn_finalize`` (forward) / ``bn_bwd_apply``
(backward); per-rank element counts are equal (``drop_last=True``,
``src/utils/semi_dataset.py:354-356``) so the global count is local * world.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch
import torch.distributed as dist

from . import config, ops


#: rehearsal switch (``SSECG_FORCE_SYNC_COLLECTIVES=1``): issue the SyncBatchNorm all-reduces even in a process group of ONE rank.
#: A one-GPU box cannot hold two RCCL ranks, but a world-size-1 ``nccl`` group runs every collective of the step through
#: ProcessGroupNCCL's real stream / event / tensor-lifetime machinery (tests/test_ddp_gpu.py::test_rccl_single_rank_rehearsal,
#: ``SSECG_BENCH_FORCE_DIST=1 python bench.py``): same code path as N > 1, results equal to the non-distributed run.
FORCE_SYNC_COLLECTIVES = config.switch("SSECG_FORCE_SYNC_COLLECTIVES", False, "rehearsal: issue the SyncBN all-reduces in a one-rank group too",
                                       __name__, "FORCE_SYNC_COLLECTIVES")


@dataclass
class BNState:
    """Parameters/buffers of one BatchNorm1d plus its mode, as plain tensors."""
    weight: torch.Tensor
    bias: torch.Tensor
    running_mean: torch.Tensor
    running_var: torch.Tensor
    num_batches_tracked: Optional[torch.Tensor]
    eps: float
    momentum: float
    group: object = None  # torch.distributed process group for SyncBN, or None

    @staticmethod
    def of(bn: torch.nn.Module, sync_ok: bool = True) -> "BNState":
        if bn.momentum is None:
            raise NotImplementedError("cumulative-average BatchNorm (momentum=None) is not on the hot path")
        group = None
        if sync_ok and isinstance(bn, torch.nn.SyncBatchNorm) and dist.is_available() and dist.is_initialized():
            g = bn.process_group if bn.process_group is not None else dist.group.WORLD
            if dist.get_world_size(g) > 1 or FORCE_SYNC_COLLECTIVES:
                group = g
        return BNState(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                       float(bn.eps), float(bn.momentum), group)


class UnitCtx:
    """What one train-mode unit keeps for its backward.  Tensors travel through
    ``ctx.save_for_backward`` (no reference cycles through the autograd node); the rest is metadata."""
    __slots__ = ("x", "w", "c", "y", "mean", "invstd", "gamma", "beta", "x_scale", "x_shift", "relu", "stride", "pad", "dil",
                 "count", "group", "aff")
    NT = 10

    def tensors(self):
        return (self.x, self.w, self.c, self.y, self.mean, self.invstd, self.gamma, self.beta, self.x_scale, self.x_shift)

    def meta(self):
        return (self.relu, self.stride, self.pad, self.dil, self.count, self.group)

    @staticmethod
    def rebuild(tensors, meta) -> "UnitCtx":
        u = UnitCtx()
        u.x, u.w, u.c, u.y, u.mean, u.invstd, u.gamma, u.beta, u.x_scale, u.x_shift = tensors
        u.aff = None
        u.relu, u.stride, u.pad, u.dil, u.count, u.group = meta
        return u


def _save_units(ctx, units, extra=()):
    """units: list of UnitCtx or None.  Saves all tensors (+ extra tensors) on the autograd ctx."""
    flat, metas = [], []
    for u in units:
        if u is None:
            metas.append(None)
        else:
            flat.extend(u.tensors())
            metas.append(u.meta())
    ctx.unit_metas = metas
    ctx.n_extra = len(extra)
    ctx.save_for_backward(*flat, *extra)


def _load_units(ctx):
    saved = ctx.saved_tensors
    units, k = [], 0
    for m in ctx.unit_metas:
        if m is None:
            units.append(None)
        else:
            units.append(UnitCtx.rebuild(saved[k:k + UnitCtx.NT], m))
            k += UnitCtx.NT
    return units, saved[k:]


#: every collective this module issues is appended here as (kind, numel, dtype) when the list is not None - the 2-rank
#: tests compare the sequences of the ranks (a mismatch in order or size is a hang or silent corruption on RCCL)
COLLECTIVE_LOG = None
#: number of SyncBatchNorm collectives issued by this process so far (the gradient reducer compares it across a backward pass)
COLLECTIVES_ISSUED = [0]


def _allreduce_sums(sums: torch.Tensor, group) -> torch.Tensor:
    COLLECTIVES_ISSUED[0] += 1
    if COLLECTIVE_LOG is not None:
        COLLECTIVE_LOG.append(("bn_sums", sums.numel(), str(sums.dtype)))
    dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
    return sums


def _allreduce_sums_async(sums: torch.Tensor, group):
    """Start the all-reduce of one BatchNorm's fp64 sums and return (tensor, work): the collective runs on the backend's
    own stream (RCCL) while the caller keeps enqueuing independent kernels; ``work.wait()`` orders the current stream
    behind it.  Collectives are still ISSUED in program order, identically on every rank."""
    COLLECTIVES_ISSUED[0] += 1
    if COLLECTIVE_LOG is not None:
        COLLECTIVE_LOG.append(("bn_sums", sums.numel(), str(sums.dtype)))
    return sums, dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group, async_op=True)


# ``num_batches_tracked += 1`` of every train-mode BatchNorm of a forward pass: collected and applied as ONE multi-tensor
# launch by ``flush_counters`` (called at the end of the backbone's and the head's forward) instead of 21 one-element adds.
_pending_counters = []


def _count_batch(nbt):
    if nbt is not None:
        _pending_counters.append(nbt)


def flush_counters():
    if _pending_counters:
        torch._foreach_add_(_pending_counters, 1)
        _pending_counters.clear()


def unit_fwd_train_begin(x, w, bn: BNState, stride, pad, dil=1, x_affine=None, sums_out=None, reduce_now=True):
    """First half of a train-mode unit: the convolution (statistics in its epilogue) and, under SyncBatchNorm, the all-reduce of
    the sums - a SYNCHRONOUS collective, which this torch launches on the CURRENT stream: the apply pass needs the result
    next, nothing can overlap it, and the asynchronous form costs two cross-stream hand-offs per BatchNorm (MEASURED on one
    rank with the collectives forced: ~20 us of idle device time each, 21 per forward, profiles/r04_dist_overhead_one_rank.txt).
    ``sums_out`` / ``reduce_now=False``: the caller reduces the sums of two independent units (a block's first convolution and
    its 1x1 downsample branch) in ONE collective - see ``unit_fwd_train_pair_begin``."""
    c, partial = ops.conv1d_fwd(x, w, stride, pad, dil, want_stats=True, in_affine=x_affine, w_cached=True)
    sums = None
    if bn.group is not None:
        sums = ops.bn_reduce_partials(partial, out=sums_out)
        if reduce_now:
            _allreduce_sums(sums, bn.group)
    return (x, w, bn, stride, pad, dil, x_affine, c, partial, sums)


def unit_fwd_train_pair_begin(x, a, b):
    """``a``, ``b`` = (w, bn, stride, pad, dil) of two units that read the same ``x``.  Under SyncBatchNorm with one process
    group their sums are reduced into neighbouring rows of one fp64 buffer and all-reduced together: one collective per
    downsample block instead of two (3 fewer per train-mode forward)."""
    (wa, bna, sa, pa, da), (wb, bnb, sb, pb, db) = a, b
    if bna.group is None or bna.group is not bnb.group:
        return unit_fwd_train_begin(x, wa, bna, sa, pa, da), unit_fwd_train_begin(x, wb, bnb, sb, pb, db)
    Ca, Cb = wa.shape[0], wb.shape[0]
    both = torch.empty((Ca + Cb, 2), device=x.device, dtype=torch.float64)
    st_a = unit_fwd_train_begin(x, wa, bna, sa, pa, da, sums_out=both[:Ca], reduce_now=False)
    st_b = unit_fwd_train_begin(x, wb, bnb, sb, pb, db, sums_out=both[Ca:], reduce_now=False)
    _allreduce_sums(both, bna.group)
    return st_a, st_b


#: a downsample block's identity = BatchNorm(conv1x1(x)) is consumed by bn2's apply pass only: that pass normalises the raw 1x1
#: output while reading it as its residual and the identity tensor is never written (round 6; bit-identical); 0 = write it as before
RESBN_IN_PLACE = config.switch("SSECG_RESBN_IN_PLACE", True, "downsample blocks: bn2's apply pass normalises the raw 1x1 output itself, the "
                               "identity tensor is not written", __name__, "RESBN_IN_PLACE")


def unit_fwd_train_finish(state, relu=True, residual=None, save=True, materialize=True, apply=True, res_bn=None):
    """``apply=False`` (a unit without ReLU or residual - the 1x1 downsample branch): statistics only, the caller's consumer applies
    this BatchNorm while reading ``c`` -> (None, ctx) with ``ctx.mean / invstd / gamma / beta`` for it.  ``res_bn``: such a
    (mean, invstd, gamma, beta) for ``residual``."""
    x, w, bn, stride, pad, dil, x_affine, c, partial, sums = state
    count = c.shape[0] * c.shape[2]
    want_aff = (bn.weight, bn.bias) if not materialize else None
    if sums is not None:
        count *= dist.get_world_size(bn.group)
        res = ops.bn_finalize(sums, count, bn.eps, bn.momentum, bn.running_mean, bn.running_var, affine_of=want_aff)
    else:
        res = ops.bn_stats_finalize(partial, count, bn.eps, bn.momentum, bn.running_mean, bn.running_var, affine_of=want_aff)
    mean, invstd = res[0], res[1]
    _count_batch(bn.num_batches_tracked)
    mask = None
    if not apply:
        assert not relu and residual is None and materialize
        y = None
    elif materialize:
        # the backward needs the ReLU mask: with a residual it cannot be recomputed from the BN input - the apply pass packs
        # it into bits (1/32 of the bytes the backward passes would otherwise read from the saved output)
        if save and relu and residual is not None and ops.bn_mask_supported(*c.shape):
            y, mask = ops.bn_apply_fwd(c, mean, invstd, bn.weight, bn.bias, residual, relu, want_mask=True, res_bn=res_bn)
        else:
            y = ops.bn_apply_fwd(c, mean, invstd, bn.weight, bn.bias, residual, relu, res_bn=res_bn)
    else:
        assert relu and residual is None
        y = None
    ctx = None
    if save:
        ctx = UnitCtx()
        # ctx.y: the mask source of a unit with a residual - packed bits (uint8) or, where those are not available, the saved
        # output; None = no residual: the mask is recomputed from the BN input c (one tensor less to read in both backward passes)
        ctx.x, ctx.w, ctx.c, ctx.y = x, w, c, ((mask if mask is not None else y) if (relu and residual is not None) else None)
        ctx.mean, ctx.invstd, ctx.gamma, ctx.beta = mean, invstd, bn.weight, bn.bias
        ctx.x_scale, ctx.x_shift = x_affine if x_affine is not None else (None, None)
        ctx.relu, ctx.stride, ctx.pad, ctx.dil = relu, stride, pad, dil
        ctx.count, ctx.group = count, bn.group
        ctx.aff = res[2] if not materialize else None
    return y, ctx


def unit_fwd_train(x, w, bn: BNState, stride, pad, dil=1, relu=True, residual=None, save=True, x_affine=None,
                   materialize=True, res_bn=None):
    """conv -> train-mode BN -> [+residual] -> [ReLU].

    ``x_affine`` = (scale, shift): ``x`` is a producer's RAW conv output and the producer's BN + ReLU is applied inside
    this conv's gather (and later inside its weight-gradient kernel).  ``materialize=False`` (needs relu, no residual):
    do not write the post-BN activation at all - return (None, ctx) with ``ctx.aff`` = this unit's (scale, shift) for
    its consumer."""
    return unit_fwd_train_finish(unit_fwd_train_begin(x, w, bn, stride, pad, dil, x_affine), relu, residual, save, materialize,
                                 res_bn=res_bn)


def unit_fwd_eval(x, w, bn: BNState, stride, pad, dil=1, relu=True, residual=None):
    scale, shift = ops.bn_fold_cached(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)
    y, _ = ops.conv1d_fwd(x, w, stride, pad, dil, scale=scale, shift=shift, residual=residual, relu=relu, w_cached=True)
    return y


# (A second HIP stream for the weight gradients - MFMA-bound next to the HBM-bound BatchNorm backward passes - measured 1.1 ms/step
# SLOWER in round 1 and was removed in round 4: tools/experiments/r04_wgrad_side_stream.patch.)
#: BasicBlock: apply bn1 + ReLU inside conv2's input staging (and its weight-gradient kernel) instead of materialising the
#: activation: 8 bn_apply passes and ~1 GB of saved activations less.  MEASURED (round 1, B=512, C=12, same box): with the
#: direct kernels the extra per-element work landed in VALU-sensitive staging phases and cancelled the gain (29.5 off vs
#: 29.7 on); in the Winograd kernels the staging VALU is free (the no-transform ablation changed nothing) and the fusion
#: gains 0.19 ms/step (24.93 -> 24.74), so it is ON by default; SSECG_FUSE_BN=0 disables it (parity-tested both ways).
FUSE_BN_INTO_CONSUMER = config.switch("SSECG_FUSE_BN", True, "bn1 + ReLU of a block applied inside conv2's gather (never written)", __name__,
                                      "FUSE_BN_INTO_CONSUMER")
#: backward of a downsample block: 1 = the 1x1 downsample branch's data gradient first and the main branch accumulates onto it
#: (rounds 1-3: a zero fill + an accumulate read per phase); default: main branch first, downsample adds in place (bit-identical in fp32)
DS_BRANCH_FIRST = config.switch("SSECG_DS_FIRST", False, "backward of a downsample block in the order of rounds 1-3", __name__, "DS_BRANCH_FIRST")


#: a downsample block's dz (the gradient behind its final ReLU) is consumed by the 1x1 branch's BatchNorm backward only: that unit masks
#: ``dout`` with the block's ReLU mask while reading it and dz is never written (round 6; bit-identical); 0 = write dz as before
DZ_IN_PLACE = config.switch("SSECG_DZ_IN_PLACE", True, "downsample blocks: the 1x1 branch's BN backward masks dout itself, dz is not written",
                            __name__, "DZ_IN_PLACE")


#: a downsample block's bn2 and the BatchNorm of its 1x1 branch receive the same masked gradient: their backward reductions and apply
#: passes run as ONE launch each (ops.bn_bwd_reduce_pair / _apply_pair: dout and the mask are read once for the two; under
#: SyncBatchNorm their sums travel in one all-reduce) - round 6, bit-identical; 0 = two launches each as before
PAIR_DS_BWD = config.switch("SSECG_PAIR_DS_BWD", True, "downsample blocks: bn2 and the 1x1 branch's BatchNorm share their backward passes",
                            __name__, "PAIR_DS_BWD")


def bn_bwd_pair(u2, ud, dout):
    """BatchNorm backward of a downsample block's bn2 (``u2``: residual + ReLU, mask in ``u2.y``) and of its 1x1 branch (``ud``) from
    the block's output gradient -> (dc2, dcd, dgamma2, dbeta2, dgamma_d, dbeta_d).  One reduction launch, one all-reduce under
    SyncBatchNorm (the two (C, 2) fp64 sums in neighbouring rows of one buffer), one apply launch."""
    C = u2.c.shape[1]
    p2, pd = ops.bn_bwd_reduce_pair(dout, u2.y, u2.c, u2.mean, u2.invstd, ud.c, ud.mean, ud.invstd)
    if u2.group is not None:
        both = torch.empty((2 * C, 2), device=dout.device, dtype=torch.float64)
        s2, dg2, db2 = ops.bn_reduce_partials(p2, want_param_grads=True, out=both[:C])
        sd, dgd, dbd = ops.bn_reduce_partials(pd, want_param_grads=True, out=both[C:])
        _allreduce_sums(both, u2.group)
    else:
        s2, dg2, db2 = ops.bn_reduce_partials(p2, want_param_grads=True)
        sd, dgd, dbd = ops.bn_reduce_partials(pd, want_param_grads=True)
    dc2, dcd = ops.bn_bwd_apply_pair(dout, u2.y, u2.c, u2.mean, u2.invstd, u2.gamma, s2, ud.c, ud.mean, ud.invstd, ud.gamma, sd, u2.count)
    return dc2, dcd, dg2, db2, dgd, dbd


def _wgrad(dc, x, k, stride, pad, dil, x_affine=None):
    return ops.conv1d_wgrad(dc, x, k, stride, pad, dil, x_affine=x_affine)


def unit_bwd(ctx: UnitCtx, dy, need_dx=True, dx_accumulate=None, need_dz=False, fill=None, defer_wgrad=False, dx_inplace=False,
             dy_mask=None):
    """-> (dx, dw, dgamma, dbeta, dz).  ``dz`` = dy masked by the ReLU = gradient of the residual input.
    (Folding the NEXT unit's bn_bwd_reduce into this unit's data-gradient epilogue was built, parity-tested and measured
    1.1 ms/step slower in round 3: tools/experiments/r04_bn_reduce_in_dgrad.patch.)
    Under SyncBatchNorm the all-reduce of [sum dz, sum dz*xhat] is started asynchronously and ``fill()`` - independent work the
    caller has pending, in practice the PREVIOUS unit's weight-gradient launch - is enqueued before the stream waits for it,
    so the collective's latency hides behind a 0.2-0.6 ms kernel instead of idling the GPU (21 of these per step).
    ``defer_wgrad``: return the weight-gradient launch as a callable (-> dw) instead of running it, for the next ``fill``.
    ``dy_mask`` (a unit WITHOUT a ReLU of its own - the 1x1 downsample branch): the incoming gradient is ``dy`` masked by another
    unit's ReLU mask (packed bits or saved activation) - the block's ``dz`` - applied while ``dy`` is read, so ``dz`` is never
    written (round 6: one 131 MB store per downsample block; the same values bit for bit)."""
    recomp = ctx.relu and ctx.y is None
    ymask = ctx.y
    if dy_mask is not None:
        if ctx.relu:
            raise ValueError("unit_bwd: dy_mask is for units without a ReLU of their own")
        ymask = dy_mask
    partial = ops.bn_bwd_reduce(dy, ymask, ctx.c, ctx.mean, ctx.invstd, ctx.gamma, ctx.beta, relu_recompute=recomp)
    # dgamma / dbeta are written from the RANK-LOCAL sums into their own tensors (DDP averages them, as PyTorch's SyncBN does);
    # the fp64 ``sums`` buffer itself is all-reduced in place (no copy) and only bn_bwd_apply reads it afterwards
    sums, dgamma, dbeta = ops.bn_reduce_partials(partial, want_param_grads=True)
    if ctx.group is not None and fill is not None:
        sums, work = _allreduce_sums_async(sums, ctx.group)
        fill()
        work.wait()
    elif ctx.group is not None:    # nothing to overlap: the synchronous form runs on this stream (no cross-stream hand-offs)
        _allreduce_sums(sums, ctx.group)
    elif fill is not None:
        fill()
    dc, dz = ops.bn_bwd_apply(dy, ymask, ctx.c, ctx.mean, ctx.invstd, ctx.gamma, sums, ctx.count, want_dz=need_dz,
                              beta=ctx.beta, relu_recompute=recomp)
    k = ctx.w.shape[2]
    x_aff = (ctx.x_scale, ctx.x_shift) if ctx.x_scale is not None else None

    def launch_wgrad():
        return _wgrad(dc, ctx.x, k, ctx.stride, ctx.pad, ctx.dil, x_affine=x_aff)

    dw = launch_wgrad if defer_wgrad else launch_wgrad()
    dx = None
    if need_dx:
        dx = ops.conv1d_dgrad(dc, ctx.w, ctx.x.shape[2], ctx.stride, ctx.pad, ctx.dil, accumulate=dx_accumulate, w_cached=True,
                              inplace=dx_inplace)
    return dx, dw, dgamma, dbeta, dz


# ----------------------------------------------------------------------------- autograd nodes
def _bn_args(bn: BNState):
    return (bn.weight, bn.bias)


class StemFn(torch.autograd.Function):
    """conv k7 s2 p3 -> BN -> ReLU -> MaxPool(k3,s2,p1)   (resnet.py:245-257, 354-355).
    The post-BN activation is never materialised: forward pools straight from the conv output, backward recomputes
    it in registers to route the pooled gradient and apply the ReLU mask (ssecg_bn_relu_maxpool_*)."""

    @staticmethod
    def forward(ctx, x, w, gamma, beta, bn: BNState, training: bool, blocked: bool = False, lp: bool = False):
        """``blocked`` (train mode under use_amp): the pooled output leaves as blocked bf16 (N, 64/8, Lp, 8) - the same values, bit for
        bit, as pooling to fp32 and converting (``amp.ToBlockedFn``), written by the pooling pass itself; the pooled gradient comes
        back in that layout and is converted once (exact) for the fp32 backward kernels.
        ``lp`` (train mode under use_amp): the convolution reads bf16-rounded x and w and stores a bf16-rounded output, its weight
        gradient reads bf16-rounded x and dc - autocast's 16-bit stem (dedicated stem kernels only; other shapes stay fp32)."""
        ctx.training = training
        ctx.blocked = False
        ctx.lp = 0
        # ``x`` may be an ops.BatchPair (labelled, unlabelled): the convolution and its weight gradient read the two tensors where they
        # lie; every other path gets the concatenation
        pair = x if isinstance(x, ops.BatchPair) else None
        if pair is not None and not (training and ops.stem_pair_ok(pair, w)):
            x, pair = pair.cat(), None
        ctx.pair = pair is not None
        if not training:
            scale, shift = ops.bn_fold_cached(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)
            if blocked and lp and ops._stem_ok(x.shape[0], x.shape[1], x.shape[2], w.shape[0], w.shape[2], 2, 3, 1):
                # eval mode INSIDE autocast (``evaluate`` under use_amp, src/algorithms/base.py:202): the 16-bit stem of the train path
                # with the running statistics - conv on bf16-rounded x and w, its output rounded (stored as bf16 where the shape
                # allows), BN + ReLU + MaxPool rounded once into the blocked layout
                x = x if x.is_contiguous() else x.contiguous()
                c, _ = ops.stem_fwd_pair(x, w, want_stats=False, lp=2 if ops.stem_c16_ok(x, None) else 1)
                if ops.stem_pool_b16_supported(*c.shape):
                    return ops.stem_pool_fwd_b16(c, None, None, scale, shift)
                from . import amp as _amp
                return _amp.to_blocked(ops.bn_relu_maxpool_fwd(c.float(), None, None, scale, shift, 3, 2, 1))
            y = ops.stem_fwd_eval_pool(x, w, scale, shift)   # one launch; the conv output is never written
            if y is None:
                c, _ = ops.conv1d_fwd(x, w, 2, 3, 1)
                y = ops.bn_relu_maxpool_fwd(c, None, None, scale, shift, 3, 2, 1)
            if blocked:      # eval under autocast with the 16-bit stem switched off (SSECG_AMP_STEM_LP=0) or a shape outside the stem kernels
                from . import amp as _amp
                y = _amp.to_blocked(y)
            return y
        # lp = 2: c (bf16-valued under lp) and dc are STORED as bf16 - where the pooling pass that reads bf16 c exists (blocked output)
        n_all = x.shape[0] if pair is None else pair[0].shape[0] + pair[1].shape[0]
        l_out = ops.conv_out_len((x if pair is None else pair[0]).shape[2], w.shape[2], 2, 3, 1)
        mode = 0
        if lp:
            # (the weight gradient has its own preconditions for bf16-stored dc - L % 4 == 0, 16-byte aligned operands -
            # that the forward's Lout % 8 test does not imply, e.g. L = 1999: ONE query answers for both entry points)
            c16 = ops.stem_c16_ok(x, None) if pair is None else ops.stem_c16_ok(pair[0], pair[1])
            mode = 2 if (blocked and c16 and ops.stem_pool_b16_supported(n_all, w.shape[0], l_out)) else 1
        if lp and pair is None and ops._stem_ok(x.shape[0], x.shape[1], x.shape[2], w.shape[0], w.shape[2], 2, 3, 1) and x.is_contiguous():
            ctx.lp = mode
            c, partial = ops.stem_fwd_pair(x, w, lp=mode)
        elif pair is not None:
            ctx.lp = mode
            c, partial = ops.stem_fwd_pair(pair, w, lp=mode)
        else:
            c, partial = ops.conv1d_fwd(x, w, 2, 3, 1, want_stats=True)
        count = c.shape[0] * c.shape[2]
        if bn.group is not None:
            sums = _allreduce_sums(ops.bn_reduce_partials(partial), bn.group)
            count *= dist.get_world_size(bn.group)
            mean, invstd = ops.bn_finalize(sums, count, bn.eps, bn.momentum, bn.running_mean, bn.running_var)
        else:
            mean, invstd = ops.bn_stats_finalize(partial, count, bn.eps, bn.momentum, bn.running_mean, bn.running_var)
        _count_batch(bn.num_batches_tracked)
        ctx.blocked = bool(blocked)
        if ctx.blocked and ops.stem_pool_b16_supported(*c.shape):
            y = ops.stem_pool_fwd_b16(c, mean, invstd, bn.weight, bn.bias)
        else:
            y = ops.bn_relu_maxpool_fwd(c, mean, invstd, bn.weight, bn.bias, 3, 2, 1)
            if ctx.blocked:
                from . import amp as _amp
                y = _amp.to_blocked(y)
        if pair is not None:
            ctx.save_for_backward(pair[0], w, c, mean, invstd, bn.weight, bn.bias, pair[1])
        else:
            ctx.save_for_backward(x, w, c, mean, invstd, bn.weight, bn.bias)
        ctx.count, ctx.group = count, bn.group
        return y

    @staticmethod
    def backward(ctx, dy):
        if not ctx.training:
            raise RuntimeError("backward through an eval-mode (BN-folded) stem is not supported")
        x, w, c, mean, invstd, gamma, beta = ctx.saved_tensors[:7]
        dy = dy.contiguous()
        if ctx.blocked:
            from . import amp as _amp
            dy = _amp.to_planar(dy)
        partial = ops.bn_relu_maxpool_bwd_reduce(dy, c, mean, invstd, gamma, beta, 3, 2, 1, lp=ctx.lp)
        sums, dg, db = ops.bn_reduce_partials(partial, want_param_grads=True)
        if ctx.group is not None:
            sums = _allreduce_sums(sums, ctx.group)
        dc = ops.bn_relu_maxpool_bwd_apply(dy, c, mean, invstd, gamma, beta, sums, ctx.count, 3, 2, 1, lp=ctx.lp)
        if ctx.pair:
            dw = ops.stem_wgrad_pair(dc, ops.BatchPair(x, ctx.saved_tensors[7]), w.shape[2], lp=ctx.lp)
            dx = None      # (a BatchPair is not a differentiable input: the data batches never require a gradient)
        elif ctx.lp and not ctx.needs_input_grad[0]:
            dw, dx = ops.stem_wgrad_pair(dc, x, w.shape[2], lp=ctx.lp), None
        else:
            if dc.dtype != torch.float32:      # (bf16-stored dc and a caller that wants the input gradient: the generic kernels read fp32)
                dc = dc.float()
            dw = _wgrad(dc, x, w.shape[2], 2, 3, 1)
            dx = ops.conv1d_dgrad(dc, w, x.shape[2], 2, 3, 1) if ctx.needs_input_grad[0] else None
        return dx, dw, dg, db, None, None, None, None


class BasicBlockFn(torch.autograd.Function):
    """BasicBlock.forward (resnet.py:55-72) with an optional 1x1 downsample branch (:287-298)."""

    @staticmethod
    def forward(ctx, x, w1, g1, b1, w2, g2, b2, wd, gd, bd, bn1: BNState, bn2: BNState, bnd, stride, dilation, training):
        has_ds = wd is not None
        if training:
            # relu(bn1(conv1(x))) is consumed by conv2 only: it is never written - conv2's gather (and later its weight
            # gradient) applies bn1 + ReLU to conv1's raw output on the fly
            fuse = FUSE_BN_INTO_CONSUMER and w2.shape[1] % 16 == 0 and w2.shape[1] <= 512 and w2.shape[0] > 32
            if has_ds:
                # both branches read x and are independent: the two convolutions, then ONE all-reduce for both BatchNorms
                s1, sd = unit_fwd_train_pair_begin(x, (w1, bn1, stride, dilation, dilation), (wd, bnd, stride, 0, 1))
                a1, u1 = unit_fwd_train_finish(s1, True, None, materialize=not fuse)
                # the identity BatchNorm(conv1x1(x)) is read by bn2's apply pass only: that pass applies it to the raw 1x1 output
                idt, ud = unit_fwd_train_finish(sd, False, None, apply=not RESBN_IN_PLACE)
            else:
                a1, u1 = unit_fwd_train(x, w1, bn1, stride, dilation, dilation, True, None, materialize=not fuse)
                idt, ud = x, None
            res_bn = None
            if has_ds and idt is None:
                idt, res_bn = ud.c, (ud.mean, ud.invstd, ud.gamma, ud.beta)
            if fuse:
                out, u2 = unit_fwd_train(u1.c, w2, bn2, 1, 1, 1, True, idt, x_affine=u1.aff, res_bn=res_bn)
            else:
                out, u2 = unit_fwd_train(a1, w2, bn2, 1, 1, 1, True, idt, res_bn=res_bn)
            _save_units(ctx, [u1, u2, ud])
        else:
            a1 = unit_fwd_eval(x, w1, bn1, stride, dilation, dilation, True, None)
            idt = unit_fwd_eval(x, wd, bnd, stride, 0, 1, False, None) if has_ds else x
            out = unit_fwd_eval(a1, w2, bn2, 1, 1, 1, True, idt)
        ctx.training, ctx.has_ds = training, has_ds
        return out

    @staticmethod
    def backward(ctx, dout):
        if not ctx.training:
            raise RuntimeError("backward through an eval-mode (BN-folded) block is not supported")
        dout = dout.contiguous()
        (u1, u2, ud), _ = _load_units(ctx)
        # each unit's weight gradient is launched inside the NEXT unit's SyncBN all-reduce window (unit_bwd: fill)
        got = {}
        # a downsample block's dz = dout * [out > 0] feeds the 1x1 branch's BatchNorm backward only: that unit masks dout itself
        # (unit_bwd: dy_mask) and dz is never written
        dz_in_place = DZ_IN_PLACE and ctx.has_ds and not DS_BRANCH_FIRST and u2.y is not None
        pending = []

        def run_pending():
            while pending:
                name, fn = pending.pop(0)
                got[name] = fn()

        if (dz_in_place and PAIR_DS_BWD and u2.group is ud.group and u2.count == ud.count and tuple(u2.c.shape) == tuple(ud.c.shape)
                and ops.bn_bwd_pair_supported(*u2.c.shape)):
            # bn2 and the 1x1 branch's BatchNorm see the same masked gradient: one reduction, one all-reduce, one apply pass for both
            dc2, dcd, dg2, db2, dgd, dbd = bn_bwd_pair(u2, ud, dout)
            aff2 = (u2.x_scale, u2.x_shift) if u2.x_scale is not None else None
            pending.append(("w2", lambda: _wgrad(dc2, u2.x, u2.w.shape[2], u2.stride, u2.pad, u2.dil, x_affine=aff2)))
            da1 = ops.conv1d_dgrad(dc2, u2.w, u2.x.shape[2], u2.stride, u2.pad, u2.dil, w_cached=True)
            dx1, w1f, dg1, db1, _ = unit_bwd(u1, da1, need_dx=True, fill=run_pending, defer_wgrad=True)
            pending.append(("w1", w1f))
            pending.append(("wd", lambda: _wgrad(dcd, ud.x, ud.w.shape[2], ud.stride, ud.pad, ud.dil)))
            dx = ops.conv1d_dgrad(dcd, ud.w, ud.x.shape[2], ud.stride, ud.pad, ud.dil, accumulate=dx1, w_cached=True, inplace=True)
            run_pending()
            return (dx, got["w1"], dg1, db1, got["w2"], dg2, db2, got["wd"], dgd, dbd, None, None, None, None, None, None)
        da1, w2, dg2, db2, dz = unit_bwd(u2, dout, need_dx=True, need_dz=not dz_in_place, defer_wgrad=True)
        pending.append(("w2", w2))

        dwd = dgd = dbd = None
        if ctx.has_ds and DS_BRANCH_FIRST:      # the order of rounds 1-3 (A/B switch SSECG_DS_FIRST=1)
            acc, wd, dgd, dbd, _ = unit_bwd(ud, dz, need_dx=True, fill=run_pending, defer_wgrad=True)
            pending.append(("wd", wd))
            dx, dw1, dg1, db1, _ = unit_bwd(u1, da1, need_dx=True, dx_accumulate=acc, fill=run_pending)
            run_pending()
            dwd = got["wd"]
        elif ctx.has_ds:
            # The main branch first: its data gradient writes EVERY position of dx (a stride-2 convolution's two parity phases), so
            # it has nothing to accumulate; the 1x1 downsample branch then adds its gradient in place (at the even positions only
            # when it has stride 2).  Rounds 1-3 ran the downsample first: a zero fill of dx, and both phases of the main branch
            # read it back (the odd phase read the zeros).  fp32 addition commutes: the result is the same, bit for bit.
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
        return dx, dw1, dg1, db1, dw2, dg2, db2, dwd, dgd, dbd, None, None, None, None, None, None


class FCNHeadFn(torch.autograd.Function):
    """conv k3 -> BN -> ReLU -> Dropout -> conv 1x1 + bias   (fcn_head.py:89-97, num_convs=1)"""

    @staticmethod
    def forward(ctx, x, w, gamma, beta, wc, bc, bn: BNState, pad, dil, drop_p, drop_mask, seed, training):
        u = None
        if training:
            a, u = unit_fwd_train(x, w, bn, 1, pad, dil, True, None)
            mask = None
            if drop_mask is not None:
                mask = drop_mask
                h = ops.mask_scale(a, mask, 1.0 / (1.0 - drop_p))
            elif drop_p > 0.0:
                h, mask = ops.dropout_fwd(a, drop_p, seed)
            else:
                h = a
            ctx.has_mask, ctx.drop_p = mask is not None, drop_p
        else:
            a = unit_fwd_eval(x, w, bn, 1, pad, dil, True, None)
            h = a
        y, _ = ops.conv1d_fwd(h, wc, 1, 0, 1, scale=None, shift=bc)
        if training:
            _save_units(ctx, [u], extra=(h, wc) + ((mask,) if mask is not None else ()))
        ctx.training = training
        ctx.has_bias = bc is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        if not ctx.training:
            raise RuntimeError("backward through an eval-mode (BN-folded) head is not supported")
        dy = dy.contiguous()
        (u,), extra = _load_units(ctx)
        h, wc = extra[0], extra[1]
        dbc = ops.channel_sum(dy) if ctx.has_bias else None
        dwc = _wgrad(dy, h, 1, 1, 0, 1)
        dh = ops.conv1d_dgrad(dy, wc, h.shape[2], 1, 0, 1)
        if ctx.has_mask:
            dh = ops.mask_scale(dh, extra[2], 1.0 / (1.0 - ctx.drop_p))
        dx, dw, dg, db, _ = unit_bwd(u, dh, need_dx=ctx.needs_input_grad[0])
        return dx, dw, dg, db, dwc, dbc, None, None, None, None, None, None, None


class ConvBNActFn(torch.autograd.Function):
    """One unit conv -> BN -> [ReLU] as its own autograd node: the building block of the FCNHead variants the shipped
    config does not use (``num_convs > 1``, ``concat_input=True``: fcn_head.py:49-80,91-93)."""

    @staticmethod
    def forward(ctx, x, w, gamma, beta, bn: BNState, stride, pad, dil, relu, training):
        ctx.training = training
        if training:
            y, u = unit_fwd_train(x, w, bn, stride, pad, dil, relu, None)
            _save_units(ctx, [u])
            return y
        return unit_fwd_eval(x, w, bn, stride, pad, dil, relu, None)

    @staticmethod
    def backward(ctx, dy):
        if not ctx.training:
            raise RuntimeError("backward through an eval-mode (BN-folded) unit is not supported")
        (u,), _ = _load_units(ctx)
        dx, dw, dg, db, _ = unit_bwd(u, dy.contiguous(), need_dx=ctx.needs_input_grad[0])
        return dx, dw, dg, db, None, None, None, None, None, None


def conv_bn_act(x, conv_weight, bn_module, stride, pad, dil, relu, training):
    return ConvBNActFn.apply(x, conv_weight, bn_module.weight, bn_module.bias, BNState.of(bn_module), stride, pad, dil, relu,
                             training)


_const_cache = {}


def _consts(device, C):
    """(zeros(C), ones(C), zeros(2C) fp64): identity BN coefficients for the standalone nodes below."""
    k = (device, C)
    if k not in _const_cache:
        _const_cache[k] = (torch.zeros(C, device=device), torch.ones(C, device=device),
                           torch.zeros(2 * C, device=device, dtype=torch.float64))
    return _const_cache[k]


class BatchNormFn(torch.autograd.Function):
    """Standalone nn.BatchNorm1d on (N, C, L) - NOT the hot path (there BN lives inside the fused units); built from the
    same kernels so that a third-party hook or head that calls a BatchNorm1d submodule directly gets the reference's
    semantics (train: batch statistics + running-stat update, SyncBN all-reduce; eval: running statistics)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, bn: BNState, training):
        x = x.contiguous()
        N, C, L = x.shape
        zeros, ones, _ = _consts(x.device, C)
        ctx.training = training
        if not training:
            scale, shift = ops.bn_fold(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)
            ctx.save_for_backward(x, scale, bn.running_mean, bn.running_var)
            ctx.eps = bn.eps
            return ops.bn_apply_fwd(x, zeros, ones, scale, shift)
        # {sum x, sum x^2} per channel: the backward reduction kernel with dy = x, mean = 0, invstd = 1
        partial = ops.bn_bwd_reduce(x, None, x, zeros, ones)
        count = N * L
        if bn.group is not None:
            sums = _allreduce_sums(ops.bn_reduce_partials(partial), bn.group)
            count *= dist.get_world_size(bn.group)
            mean, invstd = ops.bn_finalize(sums, count, bn.eps, bn.momentum, bn.running_mean, bn.running_var)
        else:
            mean, invstd = ops.bn_stats_finalize(partial, count, bn.eps, bn.momentum, bn.running_mean, bn.running_var)
        _count_batch(bn.num_batches_tracked)
        flush_counters()
        ctx.save_for_backward(x, mean, invstd, gamma)
        ctx.count, ctx.group = count, bn.group
        return ops.bn_apply_fwd(x, mean, invstd, gamma, beta)

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        if not ctx.training:
            x, scale, rmean, rvar = ctx.saved_tensors
            C = x.shape[1]
            zeros, ones, _ = _consts(x.device, C)
            invstd = torch.rsqrt(rvar + ctx.eps)
            sums, dg, db = ops.bn_reduce_partials(ops.bn_bwd_reduce(dy, None, x, rmean, invstd), want_param_grads=True)
            return ops.bn_apply_fwd(dy, zeros, ones, scale, zeros), dg, db, None, None
        x, mean, invstd, gamma = ctx.saved_tensors
        sums, dg, db = ops.bn_reduce_partials(ops.bn_bwd_reduce(dy, None, x, mean, invstd), want_param_grads=True)
        if ctx.group is not None:
            sums = _allreduce_sums(sums, ctx.group)
        dx, _ = ops.bn_bwd_apply(dy, None, x, mean, invstd, gamma, sums, ctx.count)
        return dx, dg, db, None, None


def batch_norm(x, bn_module):
    return BatchNormFn.apply(x, bn_module.weight, bn_module.bias, BNState.of(bn_module), bn_module.training)


class ReLUFn(torch.autograd.Function):
    """Standalone ReLU (any shape) through the BN kernels with identity coefficients; off the hot path (see BatchNormFn)."""

    @staticmethod
    def forward(ctx, x):
        flat = x.contiguous().view(1, 1, -1)
        zeros, ones, _ = _consts(x.device, 1)
        y = ops.bn_apply_fwd(flat, zeros, ones, ones, zeros, relu=True)
        ctx.save_for_backward(y)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        zeros, ones, zsum = _consts(dy.device, 1)
        # dz = dy * (y > 0); with zero sums and identity coefficients the BN-backward apply returns exactly dz
        dx, _ = ops.bn_bwd_apply(dy.contiguous().view(1, 1, -1), y, y, zeros, ones, ones, zsum, 1.0)
        return dx.view(dy.shape)


def relu(x):
    return ReLUFn.apply(x)



class DropoutFn(torch.autograd.Function):
    """nn.Dropout(p) in train mode (fcn_head.py:84-87,94-95): keep-mask drawn by the counter-based kernel or given."""

    @staticmethod
    def forward(ctx, x, p, mask, seed):
        if mask is not None:
            y = ops.mask_scale(x, mask, 1.0 / (1.0 - p))
        else:
            y, mask = ops.dropout_fwd(x, p, seed)
        ctx.save_for_backward(mask)
        ctx.p = p
        return y

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        return ops.mask_scale(dy.contiguous(), mask, 1.0 / (1.0 - ctx.p)), None, None, None


def dropout(x, p, mask=None, seed=0):
    return DropoutFn.apply(x, float(p), mask, int(seed))


class InterpLinearFn(torch.autograd.Function):
    """F.interpolate(mode="linear") (encoder_decoder.py:102-107)."""

    @staticmethod
    def forward(ctx, x, size, align_corners):
        ctx.in_len, ctx.align = x.shape[2], bool(align_corners)
        return ops.interp_linear_fwd(x, int(size), align_corners)

    @staticmethod
    def backward(ctx, dy):
        return ops.interp_linear_bwd(dy.contiguous(), ctx.in_len, ctx.align), None, None


def interpolate_linear(x, size, align_corners=False):
    return InterpLinearFn.apply(x, size, align_corners)


# ----------------------------------------------------------------------------- losses
class _CEHardFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, conf, thresh, denom):
        N, K, L = logits.shape
        dlogits, partial = ops.ce_hard_fwd_bwd(logits, target, conf, thresh, 1.0 / denom)
        out = ops.sum_partials(partial, 1.0 / denom)  # [mean loss, kept fraction]
        ctx.save_for_backward(dlogits)
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, gloss, _gout):
        (dlogits,) = ctx.saved_tensors
        return dlogits * gloss, None, None, None, None


def cross_entropy(logits, target):
    """F.cross_entropy(logits (N,K,L), target (N,L)) with mean reduction (fixmatch.py:105)."""
    loss, _ = _CEHardFn.apply(logits, target, None, 0.0, float(logits.shape[0] * logits.shape[2]))
    return loss


def masked_cross_entropy(logits, target, conf, thresh):
    """(F.cross_entropy(.., reduction='none') * (conf >= thresh)).mean()  (fixmatch.py:114-116).
    -> (loss, stat) with stat = [loss, mask_ratio] on device."""
    return _CEHardFn.apply(logits, target, conf, float(thresh), float(logits.shape[0] * logits.shape[2]))


class _CESoftFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, prob):
        denom = float(logits.shape[0] * logits.shape[2])
        dlogits, partial = ops.ce_soft_fwd_bwd(logits, prob, 1.0 / denom)
        out = ops.sum_partials(partial, 1.0 / denom)
        ctx.save_for_backward(dlogits)
        return out[0].clone()

    @staticmethod
    def backward(ctx, gloss):
        (dlogits,) = ctx.saved_tensors
        return dlogits * gloss, None


def soft_cross_entropy(logits, prob):
    """F.cross_entropy(logits, prob) with class-probability targets (mean_teacher.py:115)."""
    return _CESoftFn.apply(logits, prob)


class _FixMatchLossFn(torch.autograd.Function):
    """Both FixMatch terms over the concatenated student logits in one node
    (src/algorithms/fixmatch.py:102-118): rows [0,nb) supervised CE, rows [nb,N) pseudo-label CE
    weighted by (conf >= thresh); loss = (loss_x + loss_u)/2.  dlogits is written in place for both halves."""

    @staticmethod
    def forward(ctx, logits, nb, mask_x, mask_u, conf, thresh):
        N, K, L = logits.shape
        nu = N - nb
        dlogits = torch.empty_like(logits)
        _, px = ops.ce_hard_fwd_bwd(logits[:nb], mask_x, None, 0.0, 0.5 / (nb * L), dlogits=dlogits[:nb])
        _, pu = ops.ce_hard_fwd_bwd(logits[nb:], mask_u, conf, thresh, 0.5 / (nu * L), dlogits=dlogits[nb:])
        out = ops.loss_pair_finish(px, pu, 1.0 / (nb * L), 1.0 / (nu * L))   # [loss, loss, loss_x, loss_u_s, mask_ratio]
        loss, stats = out[0], out[1:5]
        ctx.save_for_backward(dlogits)
        ctx.mark_non_differentiable(stats)
        return loss, stats

    @staticmethod
    def backward(ctx, gloss, _gstats):
        (dlogits,) = ctx.saved_tensors
        return dlogits * gloss, None, None, None, None, None


def fixmatch_loss(logits, num_lb, mask_x, mask_u_w, conf_u_w, conf_thresh):
    """-> (loss, stats) with stats = [loss_total, loss_x, loss_u_s, mask_ratio] (device, non-differentiable)."""
    return _FixMatchLossFn.apply(logits, int(num_lb), mask_x, mask_u_w, conf_u_w, float(conf_thresh))


class _MeanTeacherLossFn(torch.autograd.Function):
    """src/algorithms/mean_teacher.py:103-117: hard CE on the labelled rows, soft-target CE on the rest."""

    @staticmethod
    def forward(ctx, logits, nb, mask_x, prob_u_w):
        N, K, L = logits.shape
        nu = N - nb
        dlogits = torch.empty_like(logits)
        _, px = ops.ce_hard_fwd_bwd(logits[:nb], mask_x, None, 0.0, 0.5 / (nb * L), dlogits=dlogits[:nb])
        _, pu = ops.ce_soft_fwd_bwd(logits[nb:], prob_u_w, 0.5 / (nu * L), dlogits=dlogits[nb:])
        out = ops.loss_pair_finish(px, pu, 1.0 / (nb * L), 1.0 / (nu * L))   # [loss, loss, loss_x, loss_u, -]
        loss, stats = out[0], out[1:4]
        ctx.save_for_backward(dlogits)
        ctx.mark_non_differentiable(stats)
        return loss, stats

    @staticmethod
    def backward(ctx, gloss, _gstats):
        (dlogits,) = ctx.saved_tensors
        return dlogits * gloss, None, None, None


def mean_teacher_loss(logits, num_lb, mask_x, prob_u_w):
    """-> (loss, stats) with stats = [loss_total, loss_x, loss_u_s]."""
    return _MeanTeacherLossFn.apply(logits, int(num_lb), mask_x, prob_u_w)


def pseudo_label(logits, want_prob=False):
    """conf, argmax mask[, softmax] of the teacher logits (fixmatch.py:90-91, mean_teacher.py:92)."""
    return ops.softmax_conf_argmax(logits, want_prob)


def seg_confusion(pred, target, num_classes):
    """Per-record confusion counts (N, K, K) int32 on device (rows = target class, columns = predicted class)."""
    return ops.seg_confusion(pred, target, num_classes)


def iou_from_confusion(counts, include_background=True):
    """Per-record, per-class IoU (N, K') float64 from ``seg_confusion`` counts, with the convention of
    torchmetrics 1.5.2 ``_mean_iou_update/_mean_iou_compute`` (the metric the reference validates with,
    src/utils/perf_metrics.py:9-47) and of ST++'s ``calculate_miou`` (src/algorithms/stpp.py:32-42):
    intersection / (|pred| + |target| - intersection), 0 where the union is empty."""
    c = counts.to(torch.float64)
    inter = torch.diagonal(c, dim1=1, dim2=2)
    union = c.sum(dim=2) + c.sum(dim=1) - inter
    iou = torch.where(union > 0, inter / union.clamp(min=1.0), torch.zeros_like(union))
    return iou if include_background else iou[:, 1:]
