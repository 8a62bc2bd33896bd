"""FCN decode head on the MI355X hot path (``src/models/decode_heads/fcn_head.py:9-97``).

Same constructor keywords, ``.align_corners`` attribute, state_dict keys
(``convs.0.{0,1}.*``, ``cls_seg.*``) and ``forward(tuple) -> (N, num_classes, L')``.
The shipped configuration (``num_convs: 1, concat_input: false``) is one fused HIP
node: conv k3 + BN + ReLU + dropout + 1x1 classifier.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ssecg import functional as SF
from ssecg import ops
from ssecg.nn import BatchNorm1d, Conv1d, ReLU


class FCNHead(nn.Module):
    def __init__(self, in_channels: int, channels: int, num_classes: int, num_convs: int, kernel_size: int = 3,
                 concat_input: bool = True, dilation: int = 1, in_index: int = -1, dropout_ratio: float = 0.1,
                 align_corners: bool = False, norm_layer=None, act_layer=None):
        super().__init__()
        if num_convs != 1 or concat_input:
            raise NotImplementedError("hot path covers the shipped head: num_convs=1, concat_input=False")
        assert dilation > 0
        self.num_classes, self.in_index, self.align_corners = num_classes, in_index, align_corners
        self.num_convs, self.concat_input, self.kernel_size = num_convs, concat_input, kernel_size
        self.dilation = dilation
        pad = (kernel_size // 2) * dilation
        self.convs = nn.Sequential(nn.Sequential(
            Conv1d(in_channels, channels, kernel_size, padding=pad, dilation=dilation, bias=False),
            BatchNorm1d(channels), ReLU(inplace=True)))
        self.cls_seg = Conv1d(channels, num_classes, 1)
        self.dropout_ratio = float(dropout_ratio)
        self.dropout = nn.Dropout(dropout_ratio) if dropout_ratio > 0 else None  # ratio holder; fused in the node
        #: test hook: a uint8 keep-mask (N, channels, L') used instead of drawing one (parity fixtures)
        self.fixed_dropout_mask = None

    def forward(self, inputs):
        ops.begin_forward_unless_scoped()
        x = inputs[self.in_index]
        conv, bn = self.convs[0][0], self.convs[0][1]
        p = self.dropout.p if (self.dropout is not None and self.training) else 0.0
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if (p > 0 and self.fixed_dropout_mask is None) else 0
        mask = self.fixed_dropout_mask if (self.training and p > 0) else None
        out = SF.FCNHeadFn.apply(x, conv.weight, bn.weight, bn.bias, self.cls_seg.weight, self.cls_seg.bias,
                                 SF.BNState.of(bn), conv.padding, conv.dilation, p, mask, seed, self.training)
        SF.flush_counters()
        return out
