"""FCN decode head on the MI355X hot path (``src/models/decode_heads/fcn_head.py:9-97``).

Same constructor keywords, ``.align_corners`` attribute, state_dict keys
(``convs.0.{0,1}.*``, ``cls_seg.*``) and ``forward(tuple) -> (N, num_classes, L')``.
The shipped configuration (``num_convs: 1, concat_input: false``) is one fused HIP
node: conv k3 + BN + ReLU + dropout + 1x1 classifier; the other constructor combinations
(``num_convs`` 0 or > 1, ``concat_input: true``) chain the same fused units node by node.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ssecg import functional as SF
from ssecg import ops
from ssecg.nn import BatchNorm1d, Conv1d, ReLU


class FCNHead(nn.Module):
    _ssecg_amp_capable = True
    amp = False
    amp_eval = False

    def __init__(self, in_channels: int, channels: int, num_classes: int, num_convs: int, kernel_size: int = 3,
                 concat_input: bool = True, dilation: int = 1, in_index: int = -1, dropout_ratio: float = 0.1,
                 align_corners: bool = False, norm_layer=None, act_layer=None):
        super().__init__()
        if norm_layer not in (None, nn.BatchNorm1d, BatchNorm1d) or act_layer not in (None, nn.ReLU, ReLU):
            raise NotImplementedError("only BatchNorm1d + ReLU are fused on the hot path")
        assert num_convs >= 0 and dilation > 0
        if num_convs == 0:
            assert in_channels == channels
        self.num_classes, self.in_index, self.align_corners = num_classes, in_index, align_corners
        self.num_convs, self.concat_input, self.kernel_size = num_convs, concat_input, kernel_size
        self.dilation = dilation
        pad = (kernel_size // 2) * dilation

        def unit(cin, p, d):
            return nn.Sequential(Conv1d(cin, channels, kernel_size, padding=p, dilation=d, bias=False),
                                 BatchNorm1d(channels), ReLU(inplace=True))

        if num_convs == 0:
            self.convs = nn.Identity()
        else:
            self.convs = nn.Sequential(unit(in_channels, pad, dilation), *[unit(channels, pad, dilation)
                                                                           for _ in range(num_convs - 1)])
        if concat_input:
            self.conv_cat = unit(in_channels + channels, kernel_size // 2, 1)
        self.cls_seg = Conv1d(channels, num_classes, 1)
        self.dropout_ratio = float(dropout_ratio)
        self.dropout = nn.Dropout(dropout_ratio) if dropout_ratio > 0 else None  # ratio holder; fused in the node
        #: test hook: a uint8 keep-mask (N, channels, L') used instead of drawing one (parity fixtures); a LIST of masks is
        #: consumed one per train-mode forward (gradient-accumulation fixtures: one mask per micro-step)
        self.fixed_dropout_mask = None

    def forward(self, inputs):
        ops.begin_forward_unless_scoped()
        x = inputs[self.in_index]
        p = self.dropout.p if (self.dropout is not None and self.training) else 0.0
        seed = ops.draw_seed() if (p > 0 and self.fixed_dropout_mask is None) else 0
        mask = self.fixed_dropout_mask if (self.training and p > 0) else None
        if isinstance(mask, list):
            mask = mask.pop(0)
        from ssecg import amp as SAMP
        if SAMP.is_blocked(x):
            # use_amp: the backbone hands over blocked bf16 (train mode: the student pass; eval mode under ssecg.amp.eval_autocast:
            # ``evaluate``).  The conv units of the shipped form (k3, dilation 1, no concat) run in bf16, then fp32 from the dropout
            # on; every other constructor combination (``concat_input``, ``num_convs`` 0, dilated / non-k3 units) converts to fp32
            # (exact) and runs the fp32 general form below - more precise than autocast, never an error (round 6)
            lp_units = (not self.concat_input and self.num_convs >= 1
                        and all(seq[0].dilation == 1 and seq[0].kernel_size[0] == 3 for seq in self.convs))
            if lp_units:
                out = x
                for seq in self.convs:
                    if self.training:
                        out = SAMP.UnitAmpFn.apply(out, seq[0].weight, seq[1].weight, seq[1].bias, SF.BNState.of(seq[1]), 1,
                                                   seq[0].padding, True)
                    else:
                        out = SAMP.unit_fwd_eval(out, seq[0].weight, SF.BNState.of(seq[1]), 1, seq[0].padding, True)
                out = SAMP.ToPlanarFn.apply(out)
                if p > 0:
                    out = SF.dropout(out, p, mask, seed)
                out = self.cls_seg(out)
                SF.flush_counters()
                return out
            x = SAMP.ToPlanarFn.apply(x)
        if self.num_convs == 1 and not self.concat_input:
            # the shipped head (configs/base/resnet18/*.yaml): ONE fused node conv k3 + BN + ReLU + dropout + 1x1 classifier
            conv, bn = self.convs[0][0], self.convs[0][1]
            out = SF.FCNHeadFn.apply(x, conv.weight, bn.weight, bn.bias, self.cls_seg.weight, self.cls_seg.bias,
                                     SF.BNState.of(bn), conv.padding, conv.dilation, p, mask, seed, self.training)
            SF.flush_counters()
            return out
        # general form (fcn_head.py:89-97): the same fused units chained as separate autograd nodes
        out = x
        if self.num_convs > 0:
            for seq in self.convs:
                out = SF.conv_bn_act(out, seq[0].weight, seq[1], 1, seq[0].padding, seq[0].dilation, True, self.training)
        if self.concat_input:
            cc = self.conv_cat
            out = SF.conv_bn_act(torch.cat([x, out], dim=1), cc[0].weight, cc[1], 1, cc[0].padding, cc[0].dilation, True,
                                 self.training)
        if p > 0:
            out = SF.dropout(out, p, mask, seed)
        out = self.cls_seg(out)
        SF.flush_counters()
        return out
