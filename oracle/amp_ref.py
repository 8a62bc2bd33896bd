"""ORACLE (test infrastructure, NOT product code): CPU emulation of the reduced-precision student pass.

What ``use_amp: true`` means in the reference: the student forward runs under ``torch.cuda.amp.autocast``
(``src/algorithms/fixmatch.py:97``, ``mean_teacher.py:98``, ``base.py:122``) - convolutions, BatchNorm, ReLU in
16 bit with fp32 accumulation and fp32 batch statistics, the loss in fp32, fp32 master weights; the teacher /
pseudo-label pass is outside autocast (fp32).  CUDA autocast cannot execute in the build container (no CUDA;
``torch.autocast('cpu')`` follows a different op policy), and the reference holds no fixtures for it, so this file is
an EMULATION, not a pinned restatement: **parity unpinned** for SURVEY.md row N4 (DESIGN.md says the same).

It restates, op by op, the precision policy of ``semi-seg-ecg_amd/ssecg/amp.py`` + ``csrc/amp.hip`` with plain fp32
torch ops and explicit bf16 roundings at the points where the HIP path stores a bf16 tensor:

* stem (conv k7 + BN + ReLU + max-pool) in fp32, its pooled output rounded once;
* every conv of the body / the head's conv unit: bf16-rounded weights x bf16 activations, fp32 accumulation, output
  rounded; BatchNorm statistics in fp32 FROM the rounded conv output; BN (+residual) (+ReLU) in fp32, rounded once;
* backward: every stored activation gradient is rounded once where the HIP path stores it (``_Round.backward``);
  in a downsample block the MAIN branch's input gradient is rounded before the 1x1-downsample branch's is added to it (round 4:
  it is stored bf16 by the two stride-2 phase launches, the downsample's data gradient then accumulates in place;
  ``DS_BRANCH_FIRST = True`` restores the order of rounds 2-3, the downsample branch stored first); weight / BN-parameter
  gradients stay fp32;
* dropout, the 1x1 classifier, linear interpolation and the losses in fp32.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import torch_ref as R


class _Round(torch.autograd.Function):
    """value -> bf16 -> fp32 in the forward AND in the backward (a tensor stored in bf16 and its stored gradient)."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(torch.float32)


class _RoundGrad(torch.autograd.Function):
    """identity forward, gradient rounded to bf16 (a gradient that is stored before it is accumulated elsewhere)."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(torch.float32)


def rb(x):
    return _Round.apply(x)


def wq(w):
    """bf16 operand of an fp32 master weight; the gradient reaches the master weight unrounded."""
    return w + (w.detach().to(torch.bfloat16).to(torch.float32) - w.detach())


#: accumulation dtype of the emulated convolutions.  torch.float64 gives a SECOND correct evaluation of the same policy whose
#: conv sums round differently (exact products, one rounding): tests use the distance between the two emulations as the
#: implementation-noise floor of the policy (bf16 roundings flip on 1-ulp differences and ReLU masks flip with them).
CONV_ACC = torch.float32


def _conv(x, w, stride, pad):
    if CONV_ACC is torch.float32:
        return F.conv1d(x, w, stride=stride, padding=pad)
    return F.conv1d(x.to(CONV_ACC), w.to(CONV_ACC), stride=stride, padding=pad).to(torch.float32)


#: how the batch statistics of the emulated BatchNorms are summed.  "exact" = F.batch_norm (double accumulation on the CPU);
#: "fp32_sequential" = mean and E[x^2] from running fp32 sums over the (N, L) elements of a channel, the way a kernel's
#: per-thread partial sums accumulate (error ~1e-6 relative): the second half of the "second correct evaluation" above.
STAT_MODE = "exact"


def _bn_train(sd, name, c):
    if STAT_MODE == "exact":
        return R._bn(sd, name, c, True)
    g, b = sd[name + ".weight"], sd[name + ".bias"]
    mean = c.mean(dim=(0, 2))
    var = c.var(dim=(0, 2), unbiased=False)
    with torch.no_grad():      # same values up to fp32 summation noise; the gradient flows through the exact expressions
        flat = c.detach().permute(1, 0, 2).reshape(c.shape[1], -1)
        n = flat.shape[1]
        m_seq = torch.cumsum(flat, dim=1)[:, -1] / n
        q_seq = torch.cumsum(flat * flat, dim=1)[:, -1] / n
        v_seq = (q_seq - m_seq * m_seq).clamp_min(0.0)
    mean = mean + (m_seq - mean.detach())
    var = var + (v_seq - var.detach())
    sd[name + ".num_batches_tracked"] += 1
    return (c - mean[None, :, None]) * torch.rsqrt(var + R.BN_EPS)[None, :, None] * g[None, :, None] + b[None, :, None]


def _unit(sd, conv, bn, x, stride, pad, relu=True, residual=None):
    c = rb(_conv(x, wq(sd[conv + ".weight"]), stride, pad))
    z = _bn_train(sd, bn, c)
    if residual is not None:
        z = z + residual
    return rb(F.relu(z) if relu else z)


#: which branch's input gradient of a downsample block is stored (rounded) before the other is accumulated onto it
DS_BRANCH_FIRST = False


def _basic_block(sd, p, x, stride, has_ds):
    x_main = _RoundGrad.apply(x) if (has_ds and not DS_BRANCH_FIRST) else x
    a1 = _unit(sd, p + ".conv1", p + ".bn1", x_main, stride, 1)
    if has_ds:
        idt = _unit(sd, p + ".downsample.0", p + ".downsample.1", _RoundGrad.apply(x) if DS_BRANCH_FIRST else x, stride, 0, relu=False)
    else:
        idt = x
    return _unit(sd, p + ".conv2", p + ".bn2", a1, 1, 1, relu=True, residual=idt)


def model_forward_train(sd, x, dropout_mask=None, dropout_p: float = 0.1, align_corners=False):
    """Train-mode EncoderDecoder.forward under the bf16 policy -> seg_logits (N, K, L) fp32."""
    h = F.conv1d(x, sd["backbone.stem.0.weight"], stride=2, padding=3)
    h = F.relu(R._bn(sd, "backbone.stem.1", h, True))
    h = rb(F.max_pool1d(h, kernel_size=3, stride=2, padding=1))
    for li in range(1, 5):
        h = _basic_block(sd, f"backbone.layer{li}.0", h, 1 if li == 1 else 2, li > 1)
        h = _basic_block(sd, f"backbone.layer{li}.1", h, 1, False)
    a = _unit(sd, "decode_head.convs.0.0", "decode_head.convs.0.1", h, 1, 1)
    if dropout_mask is not None:
        a = a * dropout_mask * (1.0 / (1.0 - dropout_p))
    lo = F.conv1d(a, sd["decode_head.cls_seg.weight"], sd["decode_head.cls_seg.bias"])
    return F.interpolate(lo, size=x.shape[2], mode="linear", align_corners=align_corners)


def fixmatch_step(sd, opt, batch, cfg, epoch_frac, dropout_mask=None, dropout_p=0.1):
    """torch_ref.fixmatch_step with the student pass under the bf16 policy (teacher pass fp32, outside autocast)."""
    lr = R.lr_at(epoch_frac, cfg)
    ecg_x, mask_x = batch["labeled"]["ecg"], batch["labeled"]["target"]
    ecg_u_w, ecg_u_s = batch["unlabeled"]["ecg"], batch["unlabeled"]["ecg_aug"]
    with torch.no_grad():
        pred_u_w = R.model_forward(sd, ecg_u_w, train=False)
        conf, mask = R.pseudo_label(pred_u_w)
    nb = ecg_x.shape[0]
    logits = model_forward_train(sd, torch.cat((ecg_x, ecg_u_s)), dropout_mask, dropout_p)
    loss_x, loss_u, loss, keep = R.fixmatch_losses(logits[:nb], mask_x, logits[nb:], mask, conf, cfg["conf_thresh"])
    names = R.param_names(sd)
    grads = dict(zip(names, torch.autograd.grad(loss, [sd[k] for k in names])))
    R.adamw_step(sd, grads, opt, lr, tuple(cfg.get("betas", (0.9, 0.999))), cfg.get("eps", 1e-8), cfg["weight_decay"])
    return {"lr": lr, "pred_u_w": pred_u_w, "conf": conf, "mask": mask, "keep": keep, "logits": logits.detach(),
            "loss_x": float(loss_x.detach()), "loss_u_s": float(loss_u.detach()), "loss_total": float(loss.detach()),
            "mask_ratio": float(keep.float().mean()), "grads": grads}


def supervised_step(sd, opt, batch, cfg, epoch_frac, dropout_mask=None, dropout_p=0.1):
    lr = R.lr_at(epoch_frac, cfg)
    logits = model_forward_train(sd, batch["ecg"], dropout_mask, dropout_p)
    loss = F.cross_entropy(logits, batch["target"])
    names = R.param_names(sd)
    grads = dict(zip(names, torch.autograd.grad(loss, [sd[k] for k in names])))
    R.adamw_step(sd, grads, opt, lr, tuple(cfg.get("betas", (0.9, 0.999))), cfg.get("eps", 1e-8), cfg["weight_decay"])
    return {"lr": lr, "logits": logits.detach(), "loss": float(loss.detach()), "grads": grads}
