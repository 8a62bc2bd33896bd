"""Parameter-holding modules with the reference's state_dict layout.

``Conv1d`` / ``BatchNorm1d`` own the tensors under the same names nn.Conv1d /
nn.BatchNorm1d would (``weight``, ``bias``, ``running_mean``, ``running_var``,
``num_batches_tracked``) so checkpoints interchange with the reference
(SURVEY.md §8b); the arithmetic lives in the fused units of ``ssecg.functional``,
which read these tensors directly.  ``BatchNorm1d`` subclasses nn.BatchNorm1d so
``torch.nn.SyncBatchNorm.convert_sync_batchnorm`` (src/algorithms/fixmatch.py:290-291)
converts it, which the fused units recognise as "all-reduce the statistics".
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import ops


class _Conv1dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, stride, pad, dil):
        y, _ = ops.conv1d_fwd(x, w, stride, pad, dil, scale=None, shift=b)
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, pad, dil, b is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, pad, dil, has_b = ctx.cfg
        dy = dy.contiguous()
        dx = ops.conv1d_dgrad(dy, w, x.shape[2], stride, pad, dil) if ctx.needs_input_grad[0] else None
        dw = ops.conv1d_wgrad(dy, x, w.shape[2], stride, pad, dil)  # standalone conv: same stream
        db = ops.channel_sum(dy) if has_b else None
        return dx, dw, db, None, None, None


def conv1d(x, w, b=None, stride=1, padding=0, dilation=1):
    return _Conv1dFn.apply(x, w, b, stride, padding, dilation)


class Conv1d(nn.Module):
    """nn.Conv1d's parameters and init (kaiming_uniform(a=sqrt(5)), bias U(+-1/sqrt(fan_in)));
    standalone forward = the implicit-GEMM HIP kernel."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, bias=True):
        super().__init__()
        if kernel_size not in (1, 3, 7):
            raise NotImplementedError("hot-path conv kernels are built for kernel sizes 1, 3 and 7")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding, self.dilation = (kernel_size,), stride, padding, dilation
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            bound = 1.0 / math.sqrt(in_channels * kernel_size)
            nn.init.uniform_(self.bias, -bound, bound)

    def extra_repr(self):
        return (f"{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, stride={self.stride}, "
                f"padding={self.padding}, dilation={self.dilation}, bias={self.bias is not None}")

    def forward(self, x):
        return conv1d(x, self.weight, self.bias, self.stride, self.padding, self.dilation)


class BatchNorm1d(nn.BatchNorm1d):
    """nn.BatchNorm1d's parameters/buffers.  On the hot path the fused conv+BN units read them directly; a standalone call
    (a hook, a custom head) runs ``functional.batch_norm`` - same kernels, same train / eval / SyncBN semantics."""

    def forward(self, x):
        from . import functional as SF
        if x.dim() != 3:
            raise NotImplementedError("BatchNorm1d: (N, C, L) inputs only")
        return SF.batch_norm(x, self)


class ReLU(nn.Module):
    """Keeps the reference's module indices (stem.2, convs.0.2); fused into the units on the hot path, a standalone call
    runs ``functional.relu``."""

    def __init__(self, inplace=True):
        super().__init__()

    def forward(self, x):
        from . import functional as SF
        return SF.relu(x)
