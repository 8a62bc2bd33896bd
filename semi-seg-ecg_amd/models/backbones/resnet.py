"""ResNet-1D backbone on the MI355X hot path.

Same registry entry points, constructor keywords, ``forward`` contract
(``(N, C, L) -> tuple of stage outputs``) and state_dict keys as the reference
(``src/models/backbones/resnet.py:135-376``); the arithmetic is one fused HIP
node per stem / BasicBlock (``ssecg.functional``).  Variants no shipped config
uses (Bottleneck, deep_stem, avg_down) are outside the hot path.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import torch
import torch.nn as nn

from ssecg import functional as SF
from ssecg import ops
from ssecg.nn import BatchNorm1d, Conv1d, ReLU

__all__ = ["ResNet", "resnet18", "resnet34", "resnet50", "resnet101", "resnet152"]


class BasicBlock(nn.Module):
    expansion = 1
    _ssecg_amp_capable = True
    amp = False   # ssecg.amp.enable(model): train-mode forward on bf16 storage + bf16 MFMA (use_amp: true)
    amp_eval = False   # ssecg.amp.eval_autocast(model): eval-mode forward inside autocast (``evaluate`` under use_amp)

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample: Optional[nn.Module] = None):
        super().__init__()
        self.conv1 = Conv1d(inplanes, planes, 3, stride=stride, padding=dilation, dilation=dilation, bias=False)
        self.bn1 = BatchNorm1d(planes)
        self.relu = ReLU(inplace=True)
        self.conv2 = Conv1d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = BatchNorm1d(planes)
        self.downsample = downsample
        self.stride = stride
        self.dilation = dilation

    def forward(self, x):
        from ssecg import amp as SAMP
        if SAMP.is_blocked(x):
            return SAMP.block_forward(self, x)
        ds = self.downsample
        wd = gd = bd = bnd = None
        if ds is not None:
            wd, gd, bd, bnd = ds[0].weight, ds[1].weight, ds[1].bias, SF.BNState.of(ds[1])
        return SF.BasicBlockFn.apply(
            x, self.conv1.weight, self.bn1.weight, self.bn1.bias, self.conv2.weight, self.bn2.weight, self.bn2.bias,
            wd, gd, bd, SF.BNState.of(self.bn1), SF.BNState.of(self.bn2), bnd, self.stride, self.dilation, self.training)


class ResNet(nn.Module):
    _ssecg_amp_capable = True
    amp = False
    amp_eval = False

    def __init__(self, num_leads: int, stem_channels: int = 64, base_channels: int = 64, num_stages: int = 4,
                 strides: Sequence[int] = (1, 2, 2, 2), dilations: Sequence[int] = (1, 1, 1, 1),
                 deep_stem: bool = False, avg_down: bool = False, frozen_stages: int = -1,
                 norm_layer=None, multi_grid: Optional[Sequence[int]] = None, contract_dilation: bool = False,
                 block=BasicBlock, stage_blocks: Sequence[int] = (2, 2, 2, 2), zero_init_residual: bool = False,
                 out_indices: Sequence[int] = (0, 1, 2, 3)):
        super().__init__()
        if deep_stem or avg_down:
            raise NotImplementedError("deep_stem / avg_down are not used by any shipped config; outside the hot path")
        if block is not BasicBlock:
            raise NotImplementedError("Bottleneck ResNets (resnet50+) are outside the hot path")
        if norm_layer not in (None, nn.BatchNorm1d, BatchNorm1d):
            raise NotImplementedError("only BatchNorm1d is fused on the hot path")
        assert 1 <= num_stages <= 4, "num_stages should be in [1, 4]"
        assert len(strides) == len(dilations) == num_stages, \
            "strides and dilations should be lists of the same length as num_stages"
        if frozen_stages >= 0:
            raise NotImplementedError("frozen_stages is outside the hot path")
        self.zero_init_residual = zero_init_residual
        self.out_indices = out_indices
        self.stem_channels, self.base_channels, self.num_stages = stem_channels, base_channels, num_stages
        self.strides, self.dilations = strides, dilations
        self.stage_blocks = stage_blocks[:num_stages]
        self.inplanes = stem_channels

        self.stem = nn.Sequential(Conv1d(num_leads, stem_channels, 7, stride=2, padding=3, bias=False),
                                  BatchNorm1d(stem_channels), ReLU(inplace=True))
        self.res_layers = []
        for i, num_blocks in enumerate(self.stage_blocks):
            planes = base_channels * 2 ** i
            stage_mg = multi_grid if i == len(self.stage_blocks) - 1 else None
            layer = self._make_res_layer(planes, num_blocks, strides[i], dilations[i], stage_mg, contract_dilation)
            self.inplanes = planes * block.expansion
            name = f"layer{i + 1}"
            self.add_module(name, layer)
            self.res_layers.append(name)
        self.feat_dim = block.expansion * base_channels * 2 ** (len(self.stage_blocks) - 1)
        self._reset_parameters()

    def _make_res_layer(self, planes, num_blocks, stride, dilation, multi_grid, contract_dilation):
        downsample = None
        if stride != 1 or self.inplanes != planes:
            downsample = nn.Sequential(Conv1d(self.inplanes, planes, 1, stride=stride, bias=False), BatchNorm1d(planes))
        if multi_grid is None:
            first_dilation = dilation // 2 if (dilation > 1 and contract_dilation) else dilation
        else:
            first_dilation = multi_grid[0]
        layers = [BasicBlock(self.inplanes, planes, stride, first_dilation, downsample)]
        for i in range(1, num_blocks):
            layers.append(BasicBlock(planes, planes, 1, dilation if multi_grid is None else multi_grid[i]))
        return nn.Sequential(*layers)

    def _reset_parameters(self):
        # the reference's law: conv N(0, 2/(k*Cout)), BN gamma=1 beta=0  (resnet.py:326-339)
        for m in self.modules():
            if isinstance(m, Conv1d):
                n = m.kernel_size[0] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2.0 / n))
            elif isinstance(m, nn.BatchNorm1d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
        if self.zero_init_residual:
            for m in self.modules():
                if isinstance(m, BasicBlock):
                    nn.init.constant_(m.bn2.weight, 0)

    def forward(self, x):
        ops.begin_forward_unless_scoped()
        st = self.stem
        # use_amp: the stem stays fp32 (K = 7*C); from here to the head's conv unit activations are blocked bf16.  The stem's pooling
        # pass writes that layout itself (and its backward reads the blocked gradient): ops.AMP_STEM_BLOCKED = False
        # (SSECG_AMP_STEM_BLOCKED=0) keeps the fp32 pooled tensor + the two layout passes of rounds 2-3 (same values, bit for bit)
        # ``evaluate`` under use_amp runs its eval-mode forward inside autocast too (src/algorithms/base.py:202): amp_eval, set by
        # ssecg.amp.eval_autocast for that forward only - the pseudo-label passes of the training steps stay fp32
        amp_train = bool((self.amp and self.training) or (self.amp_eval and not self.training))
        x = SF.StemFn.apply(x, st[0].weight, st[1].weight, st[1].bias, SF.BNState.of(st[1]), self.training,
                            amp_train and ops.AMP_STEM_BLOCKED, amp_train and ops.AMP_STEM_LP)
        if amp_train and not ops.AMP_STEM_BLOCKED:
            from ssecg import amp as SAMP
            x = SAMP.ToBlockedFn.apply(x)
        outs = []
        for i, name in enumerate(self.res_layers):
            x = getattr(self, name)(x)
            if i in self.out_indices:
                outs.append(x)
        SF.flush_counters()   # num_batches_tracked += 1 for all the BatchNorms above, one launch
        return tuple(outs)


def resnet18(num_leads: int, **kwargs):
    return ResNet(num_leads=num_leads, block=BasicBlock, stage_blocks=[2, 2, 2, 2], **kwargs)


def resnet34(num_leads: int, **kwargs):
    return ResNet(num_leads=num_leads, block=BasicBlock, stage_blocks=[3, 4, 6, 3], **kwargs)


def _bottleneck(*a, **k):
    raise NotImplementedError("Bottleneck ResNets (resnet50/101/152) are outside the MI355X hot path (SURVEY.md §2)")


resnet50 = resnet101 = resnet152 = _bottleneck
