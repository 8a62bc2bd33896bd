"""Segmentor wrapper (``src/models/encoder_decoder.py:11-136``): backbone -> decode head ->
linear upsample to the input length [-> cross-entropy], dict in / dict out."""
from __future__ import annotations

from typing import Optional

import torch.nn as nn
from torch import Tensor

from ssecg import functional as SF
from ssecg import ops


class CrossEntropyLoss(nn.Module):
    """nn.CrossEntropyLoss() (mean over N*L) on (N, K, L) logits / (N, L) int64 labels, fused fwd+bwd HIP kernel."""

    def forward(self, logits, labels):
        return SF.cross_entropy(logits, labels)


class EncoderDecoder(nn.Module):
    def __init__(self, backbone: nn.Module, decode_head: nn.Module, decode_head_loss: Optional[nn.Module] = None,
                 auxiliary_heads: Optional[nn.ModuleList] = None, auxiliary_head_losses: Optional[nn.ModuleList] = None,
                 use_latent_projection: bool = False, projection_in_dim: Optional[int] = None,
                 projection_out_dim: Optional[int] = None):
        super().__init__()
        if auxiliary_heads is not None or auxiliary_head_losses is not None:
            raise NotImplementedError("auxiliary heads are dead code in the reference (SURVEY.md Q6); not on the hot path")
        if use_latent_projection:
            raise NotImplementedError("latent projection (ReCo only) is outside the hot path")
        self.backbone = backbone
        self.decode_head = decode_head
        self.loss_decode = decode_head_loss

    @property
    def with_auxiliary_heads(self) -> bool:
        return False

    @property
    def with_decode_head(self) -> bool:
        return self.decode_head is not None

    @property
    def with_projection(self) -> bool:
        return False

    def no_weight_decay(self):
        rst = set()
        if hasattr(self.backbone, "no_weight_decay"):
            rst = rst.union(self.backbone.no_weight_decay())
        if hasattr(self.decode_head, "no_weight_decay"):
            rst = rst.union(self.decode_head.no_weight_decay())
        return rst

    def forward(self, inputs: Tensor, labels: Optional[Tensor] = None, return_loss: bool = False,
                return_latent: bool = False) -> dict:
        outputs = dict()
        if isinstance(inputs, (tuple, list)) and not isinstance(inputs, ops.BatchPair):
            inputs = ops.BatchPair(*inputs)     # (a wrapper that moved / rebuilt the argument: torch's DistributedDataParallel)
        seq_len = inputs.size()[2]
        with ops.model_scope():   # weights may have been rewritten since the last forward: ONE operand refresh per forward
            x = self.backbone(inputs)
            head_out = self.decode_head(x)
        if return_latent:
            latent = x[-1]
            from ssecg import amp as SAMP
            if SAMP.is_blocked(latent):   # use_amp train mode hands over blocked bf16: back to fp32 (N, C, L) (exact)
                latent = SAMP.ToPlanarFn.apply(latent)
            outputs["latent"] = SF.interpolate_linear(latent, seq_len, self.decode_head.align_corners)
        seg_logits = SF.interpolate_linear(head_out, seq_len, self.decode_head.align_corners)
        outputs["seg_logits"] = seg_logits
        if return_loss:
            outputs["loss"] = self.loss_decode(seg_logits, labels)
        return outputs
