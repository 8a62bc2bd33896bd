"""ORACLE (test infrastructure, NOT product code).

CPU restatement, in plain ``torch.nn.functional`` fp32 ops over a flat dict of
tensors, of the reference's training hot path.  It is imported
only by ``tests/``, ``tools/make_golden.py``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg - never by the product package
(``semi-seg-ecg_amd/``), which must fail loudly without its HIP library.

Pinning: ``tools/make_golden.py`` runs this file against the reference imported
from ``/root/reference`` in the build container (models unmodified, the
algorithms' real ``train_one_epoch`` with in-memory stubs for the absent
tensorboard/torchmetrics/torch._six) and freezes the reference's outputs under
``tests/golden/``; ``tests/test_oracle_golden.py`` re-checks this restatement
against those frozen vectors on every run.  The reference itself ships no
tests or golden vectors for this path.

Every function cites the reference file:line it restates (paths relative to
``/root/reference``).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

BN_EPS = 1e-5        # nn.BatchNorm1d default, src/models/backbones/resnet.py:28
BN_MOMENTUM = 0.1


def state_from_numpy(sd_np, requires_grad: bool = True, dtype=torch.float32):
    """numpy state (ssecg.synth.model_state) -> OrderedDict of torch CPU tensors.
    ``dtype=torch.float64`` gives the fp64 "truth" used to measure the fp32 reference's own noise."""
    sd = OrderedDict()
    for k, v in sd_np.items():
        t = torch.from_numpy(v.copy())
        if t.is_floating_point():
            t = t.to(dtype)
        if requires_grad and t.is_floating_point() and not (
                k.endswith("running_mean") or k.endswith("running_var")):
            t.requires_grad_(True)
        sd[k] = t
    return sd


def _bn(sd, name, x, train: bool):
    """nn.BatchNorm1d forward: batch statistics + running-stat update in train
    mode, running statistics in eval mode (src/models/backbones/resnet.py:58,62)."""
    rm, rv = sd[name + ".running_mean"], sd[name + ".running_var"]
    if train:
        sd[name + ".num_batches_tracked"] += 1
    return F.batch_norm(x, rm, rv, sd[name + ".weight"], sd[name + ".bias"],
                        training=train, momentum=BN_MOMENTUM, eps=BN_EPS)


def _basic_block(sd, p, x, stride, has_ds, train):
    """BasicBlock.forward, src/models/backbones/resnet.py:55-72."""
    out = F.conv1d(x, sd[p + ".conv1.weight"], stride=stride, padding=1)
    out = F.relu(_bn(sd, p + ".bn1", out, train))
    out = F.conv1d(out, sd[p + ".conv2.weight"], padding=1)
    out = _bn(sd, p + ".bn2", out, train)
    if has_ds:
        idt = F.conv1d(x, sd[p + ".downsample.0.weight"], stride=stride)
        idt = _bn(sd, p + ".downsample.1", idt, train)
    else:
        idt = x
    return F.relu(out + idt)


def backbone_forward(sd, x, train: bool):
    """ResNet.forward for resnet18, src/models/backbones/resnet.py:353-363, stem :245-257."""
    x = F.conv1d(x, sd["backbone.stem.0.weight"], stride=2, padding=3)
    x = F.relu(_bn(sd, "backbone.stem.1", x, train))
    x = F.max_pool1d(x, kernel_size=3, stride=2, padding=1)
    outs = []
    for li in range(1, 5):
        stride = 1 if li == 1 else 2
        x = _basic_block(sd, f"backbone.layer{li}.0", x, stride, li > 1, train)
        x = _basic_block(sd, f"backbone.layer{li}.1", x, 1, False, train)
        outs.append(x)
    return tuple(outs)


def head_forward(sd, feats, train: bool, dropout_mask=None, dropout_p: float = 0.1):
    """FCNHead.forward (num_convs=1, concat_input=False), src/models/decode_heads/fcn_head.py:89-97.

    ``dropout_mask``: optional explicit keep-mask (N,128,L') of 0/1 so that the
    RNG-dependent nn.Dropout (fcn_head.py:84-87,94-95) is reproducible; None in
    train mode means "no dropout" (dropout_ratio 0)."""
    x = feats[3]
    x = F.conv1d(x, sd["decode_head.convs.0.0.weight"], padding=1)
    x = F.relu(_bn(sd, "decode_head.convs.0.1", x, train))
    if train and dropout_mask is not None:
        x = x * dropout_mask * (1.0 / (1.0 - dropout_p))
    return F.conv1d(x, sd["decode_head.cls_seg.weight"], sd["decode_head.cls_seg.bias"])


def model_forward(sd, x, train: bool, dropout_mask=None, dropout_p: float = 0.1, align_corners=False):
    """EncoderDecoder.forward -> seg_logits (N, num_classes, L), src/models/encoder_decoder.py:86-108."""
    feats = backbone_forward(sd, x, train)
    lo = head_forward(sd, feats, train, dropout_mask, dropout_p)
    return F.interpolate(lo, size=x.shape[2], mode="linear", align_corners=align_corners)


def pseudo_label(logits_w):
    """src/algorithms/fixmatch.py:90-91."""
    conf = logits_w.softmax(dim=1).max(dim=1)[0]
    mask = logits_w.argmax(dim=1)
    return conf, mask


def fixmatch_losses(pred_x, mask_x, pred_u_s, mask_u_w, conf_u_w, conf_thresh):
    """src/algorithms/fixmatch.py:105,114-118 (mean over ALL B*L, not over kept)."""
    loss_x = F.cross_entropy(pred_x, mask_x)
    keep = conf_u_w >= conf_thresh
    loss_u = (F.cross_entropy(pred_u_s, mask_u_w, reduction="none") * keep).mean()
    return loss_x, loss_u, (loss_x + loss_u) / 2.0, keep


def mean_teacher_losses(pred_x, mask_x, pred_u_s, prob_u_w):
    """src/algorithms/mean_teacher.py:106,115-117 (soft-target cross entropy)."""
    loss_x = F.cross_entropy(pred_x, mask_x)
    loss_u = F.cross_entropy(pred_u_s, prob_u_w)
    return loss_x, loss_u, (loss_x + loss_u) / 2.0


def lr_at(epoch_frac: float, cfg: dict) -> float:
    """src/utils/lr_sched.py:6-18."""
    if epoch_frac < cfg["warmup_epochs"]:
        return cfg["lr"] * epoch_frac / cfg["warmup_epochs"]
    return cfg["min_lr"] + (cfg["lr"] - cfg["min_lr"]) * 0.5 * (
        1.0 + math.cos(math.pi * (epoch_frac - cfg["warmup_epochs"]) / (cfg["epochs"] - cfg["warmup_epochs"])))


def param_names(sd):
    return [k for k, v in sd.items() if v.is_floating_point() and not (
        k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked"))]


def buffer_names(sd):
    p = set(param_names(sd))
    return [k for k in sd if k not in p]


def adamw_step(sd, grads, opt, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05):
    """torch.optim.AdamW single-tensor update (src/utils/optimizer.py:27-37: one
    param group, decay on every parameter incl. BN affine and cls bias)."""
    opt["step"] = opt.get("step", 0) + 1
    t = opt["step"]
    b1, b2 = betas
    bc1 = 1.0 - b1 ** t
    bc2 = 1.0 - b2 ** t
    with torch.no_grad():
        for k in param_names(sd):
            p, g = sd[k], grads[k]
            m = opt.setdefault("exp_avg." + k, torch.zeros_like(p))
            v = opt.setdefault("exp_avg_sq." + k, torch.zeros_like(p))
            p.mul_(1.0 - lr * weight_decay)
            m.lerp_(g, 1.0 - b1)
            v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
            denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
            p.addcdiv_(m, denom, value=-(lr / bc1))


def ema_update(student, teacher, decay):
    """src/algorithms/mean_teacher.py:138-149 - every parameter AND every buffer,
    including int64 num_batches_tracked, which thereby turns float32."""
    with torch.no_grad():
        for k in student:
            teacher[k] = teacher[k].detach() * decay + student[k].detach() * (1.0 - decay)


def fixmatch_step(sd, opt, batch, cfg, epoch_frac, dropout_mask=None, dropout_p=0.1):
    """One iteration of src/algorithms/fixmatch.py:73-138 (accum_iter=1, max_norm None,
    fp32, GradScaler = identity on CPU).  Returns a dict of everything observable."""
    lr = lr_at(epoch_frac, cfg)
    ecg_x, mask_x = batch["labeled"]["ecg"], batch["labeled"]["target"]
    ecg_u_w, ecg_u_s = batch["unlabeled"]["ecg"], batch["unlabeled"]["ecg_aug"]
    with torch.no_grad():
        pred_u_w = model_forward(sd, ecg_u_w, train=False)
        conf, mask = pseudo_label(pred_u_w)
    nb = ecg_x.shape[0]
    logits = model_forward(sd, torch.cat((ecg_x, ecg_u_s)), train=True,
                           dropout_mask=dropout_mask, dropout_p=dropout_p)
    pred_x, pred_u_s = logits[:nb], logits[nb:]
    loss_x, loss_u, loss, keep = fixmatch_losses(pred_x, mask_x, pred_u_s, mask, conf, cfg["conf_thresh"])
    names = param_names(sd)
    gl = torch.autograd.grad(loss, [sd[k] for k in names])
    grads = dict(zip(names, gl))
    adamw_step(sd, grads, opt, lr, tuple(cfg.get("betas", (0.9, 0.999))), cfg.get("eps", 1e-8), cfg["weight_decay"])
    return {"lr": lr, "pred_u_w": pred_u_w, "conf": conf, "mask": mask, "keep": keep,
            "logits": logits.detach(), "loss_x": float(loss_x.detach()), "loss_u_s": float(loss_u.detach()),
            "loss_total": float(loss.detach()), "mask_ratio": float(keep.float().mean()), "grads": grads}


def grad_norm(grads):
    """``get_grad_norm_`` (src/utils/misc.py:265-278, norm_type 2): the norm of the per-tensor norms."""
    return torch.norm(torch.stack([torch.norm(g.detach(), 2.0) for g in grads.values()]), 2.0)


def fixmatch_accum_step(sd, opt, batches, cfg, epoch, dropout_masks, dropout_p=0.1):
    """``accum_iter = len(batches)`` iterations of src/algorithms/fixmatch.py:73-138 ending in ONE optimiser step, as the
    reference runs them when the loaders hold exactly these batches: the lr of the first iteration (``data_iter_step /
    num_steps + epoch`` at data_iter_step 0, :73-78), every loss divided by accum_iter before its backward (:129), the
    gradients summed in ``.grad`` (misc.py:243), then ``clip_grad_norm_(max_norm)`` (misc.py:244-248; torch's own function,
    as the reference calls it) or ``get_grad_norm_`` when max_norm is None (:250-251), AdamW, zero_grad.  BN running
    statistics move at every train-mode forward, so the second micro-step's pseudo-labels see the first one's update."""
    accum = len(batches)
    assert accum == int(cfg.get("accum_iter", 1))
    lr = lr_at(float(epoch), cfg)
    names = param_names(sd)
    acc, micro = None, []
    for m, batch in enumerate(batches):
        ecg_x, mask_x = batch["labeled"]["ecg"], batch["labeled"]["target"]
        ecg_u_w, ecg_u_s = batch["unlabeled"]["ecg"], batch["unlabeled"]["ecg_aug"]
        with torch.no_grad():
            pred_u_w = model_forward(sd, ecg_u_w, train=False)
            conf, mask = pseudo_label(pred_u_w)
        nb = ecg_x.shape[0]
        logits = model_forward(sd, torch.cat((ecg_x, ecg_u_s)), train=True, dropout_mask=dropout_masks[m], dropout_p=dropout_p)
        loss_x, loss_u, loss, keep = fixmatch_losses(logits[:nb], mask_x, logits[nb:], mask, conf, cfg["conf_thresh"])
        gl = torch.autograd.grad(loss / accum, [sd[k] for k in names])
        g = dict(zip(names, gl))
        if acc is None:
            acc = {k: v.clone() for k, v in g.items()}
        else:
            for k in names:
                acc[k] += g[k]
        micro.append({"pred_u_w": pred_u_w, "conf": conf, "mask": mask, "keep": keep, "logits": logits.detach(),
                      "loss_x": float(loss_x.detach()), "loss_u_s": float(loss_u.detach()), "loss_total": float(loss.detach()),
                      "mask_ratio": float(keep.float().mean()), "grads": g})
    max_norm = cfg.get("max_norm", None)
    if max_norm is not None:
        ps = [sd[k] for k in names]
        for k in names:
            sd[k].grad = acc[k]
        norm = torch.nn.utils.clip_grad_norm_(ps, max_norm)     # scales .grad in place by min(1, max_norm / (norm + 1e-6))
        clipped = {k: sd[k].grad for k in names}
        for k in names:
            sd[k].grad = None
    else:
        norm, clipped = grad_norm(acc), acc
    adamw_step(sd, clipped, opt, lr, tuple(cfg.get("betas", (0.9, 0.999))), cfg.get("eps", 1e-8), cfg["weight_decay"])
    return {"lr": lr, "micro": micro, "norm": float(norm), "grads": clipped}


def mean_teacher_step(student, teacher, opt, batch, cfg, epoch_frac, dropout_mask=None, dropout_p=0.1):
    """One iteration of src/algorithms/mean_teacher.py:76-149."""
    lr = lr_at(epoch_frac, cfg)
    ecg_x, mask_x = batch["labeled"]["ecg"], batch["labeled"]["target"]
    ecg_u_w, ecg_u_s = batch["unlabeled"]["ecg"], batch["unlabeled"]["ecg_aug"]
    with torch.no_grad():
        pred_u_w = model_forward(teacher, ecg_u_w, train=False)
        prob = pred_u_w.softmax(dim=1)
    nb = ecg_x.shape[0]
    logits = model_forward(student, torch.cat((ecg_x, ecg_u_s)), train=True,
                           dropout_mask=dropout_mask, dropout_p=dropout_p)
    loss_x, loss_u, loss = mean_teacher_losses(logits[:nb], mask_x, logits[nb:], prob)
    names = param_names(student)
    gl = torch.autograd.grad(loss, [student[k] for k in names])
    grads = dict(zip(names, gl))
    adamw_step(student, grads, opt, lr, tuple(cfg.get("betas", (0.9, 0.999))), cfg.get("eps", 1e-8), cfg["weight_decay"])
    ema_update(student, teacher, cfg.get("ema_decay", 0.999))
    return {"lr": lr, "pred_u_w": pred_u_w, "prob": prob, "logits": logits.detach(),
            "loss_x": float(loss_x.detach()), "loss_u_s": float(loss_u.detach()), "loss_total": float(loss.detach()), "grads": grads}


def supervised_step(sd, opt, batch, cfg, epoch_frac, dropout_mask=None, dropout_p=0.1):
    """One iteration of src/algorithms/base.py:113-148 (CE inside the module,
    src/models/encoder_decoder.py:110-111)."""
    lr = lr_at(epoch_frac, cfg)
    logits = model_forward(sd, batch["ecg"], train=True, dropout_mask=dropout_mask, dropout_p=dropout_p)
    loss = F.cross_entropy(logits, batch["target"])
    names = param_names(sd)
    gl = torch.autograd.grad(loss, [sd[k] for k in names])
    grads = dict(zip(names, gl))
    adamw_step(sd, grads, opt, lr, tuple(cfg.get("betas", (0.9, 0.999))), cfg.get("eps", 1e-8), cfg["weight_decay"])
    return {"lr": lr, "logits": logits.detach(), "loss": float(loss.detach()), "grads": grads}


def _hard_pair_step(student, opt, mask_u, ecg_x, mask_x, ecg_u_w, cfg, lr, dropout_mask, dropout_p):
    """Shared tail of CPS / ST++: train pass over cat(ecg_x, ecg_u_w), loss = (CE_x + CE_u) / 2 with hard labels and no
    confidence mask (src/algorithms/cps.py:113-137, src/algorithms/stpp.py:159-183), then AdamW."""
    nb = ecg_x.shape[0]
    logits = model_forward(student, torch.cat((ecg_x, ecg_u_w)), train=True, dropout_mask=dropout_mask, dropout_p=dropout_p)
    loss_x = F.cross_entropy(logits[:nb], mask_x)
    loss_u = F.cross_entropy(logits[nb:], mask_u)
    loss = (loss_x + loss_u) / 2.0
    names = param_names(student)
    gl = torch.autograd.grad(loss, [student[k] for k in names])
    grads = dict(zip(names, gl))
    adamw_step(student, grads, opt, lr, tuple(cfg.get("betas", (0.9, 0.999))), cfg.get("eps", 1e-8), cfg["weight_decay"])
    return {"logits": logits.detach(), "loss_x": float(loss_x.detach()), "loss_u_s": float(loss_u.detach()),
            "loss_total": float(loss.detach()), "grads": grads}


def cps_step(sd1, sd2, opt1, opt2, batch, cfg, epoch_frac, dropout_masks=(None, None), dropout_p=0.1):
    """One iteration of src/algorithms/cps.py:76-170: both models label the weak view in eval mode FIRST, then model 1
    trains on model 2's labels and model 2 on model 1's.  Reported losses are the means over the two models."""
    lr = lr_at(epoch_frac, cfg)
    ecg_x, mask_x = batch["labeled"]["ecg"], batch["labeled"]["target"]
    ecg_u_w = batch["unlabeled"]["ecg"]
    with torch.no_grad():
        pred1 = model_forward(sd1, ecg_u_w, train=False)
        pred2 = model_forward(sd2, ecg_u_w, train=False)
        mask1, mask2 = pred1.argmax(dim=1), pred2.argmax(dim=1)
    r1 = _hard_pair_step(sd1, opt1, mask2, ecg_x, mask_x, ecg_u_w, cfg, lr, dropout_masks[0], dropout_p)
    r2 = _hard_pair_step(sd2, opt2, mask1, ecg_x, mask_x, ecg_u_w, cfg, lr, dropout_masks[1], dropout_p)
    out = {"lr": lr, "pred_u_w_1": pred1, "pred_u_w_2": pred2, "mask_1": mask1, "mask_2": mask2, "m1": r1, "m2": r2}
    for k in ("loss_total", "loss_x", "loss_u_s"):
        out[k] = (r1[k] + r2[k]) / 2
    return out


def stpp_step(student, teacher, opt, batch, cfg, epoch_frac, dropout_mask=None, dropout_p=0.1):
    """One iteration of src/algorithms/stpp.py:136-200 (frozen teacher in eval mode, hard labels, weak view only)."""
    lr = lr_at(epoch_frac, cfg)
    ecg_x, mask_x = batch["labeled"]["ecg"], batch["labeled"]["target"]
    ecg_u_w = batch["unlabeled"]["ecg"]
    with torch.no_grad():
        pred = model_forward(teacher, ecg_u_w, train=False)
        mask = pred.argmax(dim=1)
    r = _hard_pair_step(student, opt, mask, ecg_x, mask_x, ecg_u_w, cfg, lr, dropout_mask, dropout_p)
    r.update({"lr": lr, "pred_u_w": pred, "mask": mask})
    return r
