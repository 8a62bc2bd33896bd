"""GPU: window lengths other than the benchmark's 2000 - the reference's own configs use signal_length 2500
(configs/base/resnet18/fixmatch.yaml:47) and records need not be even.  One FixMatch step (eval pass, train pass, losses,
backward, AdamW) of the HIP path against the oracle run on the same inputs."""
import numpy as np
import pytest
import torch

from helpers import TRAIN_CFG, build_hip_model, cpu_batch, sharpen_for, to_dev
from oracle import torch_ref as O
from ssecg import functional as SF
from ssecg import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("L,C,B", [(2500, 1, 2), (1999, 2, 3), (2501, 12, 2), (250, 1, 5), (37, 2, 4)])
def test_fixmatch_step_at_other_lengths(L, C, B, dev):
    import algorithms.fixmatch as A_fm
    from utils.misc import NativeScalerWithGradNormCount
    from utils.optimizer import get_optimizer_from_config
    seed = 900 + L
    sd_np = synth.model_state(seed, C, trained=True, sharpen=sharpen_for(C))
    batch_np = synth.fixmatch_batch(seed + 1, B, C, L)
    head_len = None
    model = build_hip_model(C, sd_np, dev)
    # length of the head's feature map (stem k7 s2 p3, pool k3 s2 p1, three stride-2 stages) for the dropout mask
    l = L
    for k, s, p in ((7, 2, 3), (3, 2, 1), (3, 2, 1), (3, 2, 1), (3, 2, 1)):
        l = (l + 2 * p - k) // s + 1
    head_len = l
    dm = (synth.uniform(seed, 77, 2 * B * 128 * head_len).reshape(2 * B, 128, head_len) >= 0.1)
    model.decode_head.fixed_dropout_mask = torch.from_numpy(dm).to(dev, torch.uint8)
    cfg = dict(TRAIN_CFG)
    opt = get_optimizer_from_config(cfg, model.parameters())
    cap = []
    model.register_forward_hook(lambda m, i, o: cap.append(o["seg_logits"].detach().clone()))
    b = to_dev(batch_np, dev)
    stats = A_fm.train_one_epoch(model, [b["labeled"]], [b["unlabeled"]], opt, dev, 3, NativeScalerWithGradNormCount(), None,
                                 False, cfg)
    sd = O.state_from_numpy(sd_np)
    r = O.fixmatch_step(sd, {}, cpu_batch(batch_np), dict(TRAIN_CFG, betas=(0.9, 0.999)), 3, torch.from_numpy(dm.astype(np.float32)))
    pred_u_w, logits = cap
    assert tuple(logits.shape) == (2 * B, 4, L)
    rel = lambda a, ref: ((a.detach().double().cpu() - ref.double()).abs().max() / (ref.double().abs().max() + 1e-30)).item()
    assert rel(pred_u_w, r["pred_u_w"]) < 1e-4 and rel(logits, r["logits"]) < 1e-4
    top2 = r["pred_u_w"].topk(2, dim=1)[0]
    sure = ((top2[:, 0] - top2[:, 1]) > 1e-4).numpy()
    assert np.array_equal(SF.pseudo_label(pred_u_w)[1].cpu().numpy()[sure], r["mask"].numpy()[sure])
    for k in ("loss_total", "loss_x", "loss_u_s"):
        assert abs(stats[k] - r[k]) < 1e-4 * max(abs(r[k]), 1e-3), (k, stats[k], r[k])
    assert abs(stats["mask_ratio"] - r["mask_ratio"]) < 2e-3
    # first AdamW step: the update is lr * g / (|g| + eps) - elements whose gradient is rounding noise may land 2*lr apart;
    # everything else must agree, and nothing may move by more than ~2 updates
    msd = model.state_dict()
    for k, _ in model.named_parameters():
        d = (msd[k].detach().cpu().double() - sd[k].detach().double()).abs()
        assert d.max().item() <= 2.2 * r["lr"], k
        assert (d > 1e-5).double().mean().item() <= max(0.02, 2.0 / d.numel()), f"{k}: {(d > 1e-5).double().mean().item():.3f} of the elements differ"


@pytest.mark.parametrize("L,C,B", [(1999, 2, 3), (4095, 1, 2), (2500, 12, 2), (2002, 2, 2)])
def test_fixmatch_step_under_use_amp_at_other_lengths(L, C, B, dev):
    """under ``use_amp`` the stem stores its conv output / that output's gradient as bf16 (mode 2) only where BOTH the
    forward and the weight-gradient entry points take it.  L = 16m - 1 (1999, 4095: Lout = 8m) passes the forward's Lout % 8 test
    alone - the backward used to raise SsecgError in the middle of the step.  One FixMatch step under ``use_amp`` at such lengths
    (and at a mode-2 length, 2500 -> no, Lout 1250 % 8 != 0; 2002 -> Lout 1001): finite losses within 5 % of the fp32 step's, every
    parameter moved."""
    import copy
    import algorithms.fixmatch as A_fm
    from ssecg import ops
    from utils.misc import NativeScalerWithGradNormCount
    from utils.optimizer import get_optimizer_from_config
    seed = 1900 + L
    # (un-sharpened classifier: the 12-lead state of the fp32 tests scales its logits by 24, and a 16-bit pass's 2e-2 relative noise with them)
    model = build_hip_model(C, synth.model_state(seed, C, trained=True, sharpen=1.0), dev)
    twin = copy.deepcopy(model)
    b = to_dev(synth.fixmatch_batch(seed + 1, B, C, L), dev)
    x = b["labeled"]["ecg"]
    assert ops.stem_c16_ok(x, b["unlabeled"]["ecg_aug"]) == (L % 4 == 0 and ((L - 1) // 2 + 1) % 8 == 0)
    cfg = dict(TRAIN_CFG)
    before = {k: v.detach().clone() for k, v in model.named_parameters()}
    st = A_fm.train_one_epoch(model, [b["labeled"]], [b["unlabeled"]], get_optimizer_from_config(cfg, model.parameters()), dev, 3,
                              NativeScalerWithGradNormCount(), None, True, cfg)
    st32 = A_fm.train_one_epoch(twin, [b["labeled"]], [b["unlabeled"]], get_optimizer_from_config(cfg, twin.parameters()), dev, 3,
                                NativeScalerWithGradNormCount(), None, False, cfg)
    for k in ("loss_total", "loss_x"):
        assert np.isfinite(st[k]) and abs(st[k] - st32[k]) < 5e-2 * abs(st32[k]), (k, st[k], st32[k])
    for k, p in model.named_parameters():
        assert torch.isfinite(p).all() and not torch.equal(p.detach(), before[k]), k
