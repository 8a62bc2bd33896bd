"""GPU: the reduced-precision path (``use_amp: true``): bf16 blocked storage + bf16 MFMA kernels of
csrc/amp.hip against torch fp32 arithmetic on bf16-rounded operands (kernel level), and the whole student step against the
CPU emulation ``oracle/amp_ref.py`` (model level).  The reference's CUDA autocast cannot run without CUDA and the reference
holds no fixtures for it: PARITY UNPINNED for this row; the bars below are against the emulation."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import TRAIN_CFG, build_hip_model, cpu_batch, dropout_mask_np, rel, to_dev
from ssecg import amp as SAMP
from ssecg import functional as SF
from ssecg import ops, synth

pytestmark = pytest.mark.gpu


def rb(t):
    return t.to(torch.bfloat16).to(torch.float32)


def rnd(seed, *shape, std=1.0):
    return torch.from_numpy(synth.normal(seed, 9, shape, std=std))


def blocked(t, dev):
    return SAMP.to_blocked(t.to(dev))


def assert_bf16_close(got, ref, what, frac=0.02):
    """``got`` (fp32 view of a bf16 result) vs the fp32 reference rounded to bf16: identical up to 1 bf16 ulp on the few
    elements whose fp32 value sits at a rounding boundary (different summation order)."""
    got, ref = got.detach().float().cpu(), rb(ref.detach().float().cpu())
    d = (got - ref).abs()
    tol = ref.abs() * 2.0 ** -7 + 1e-30 + ref.abs().max() * 1e-6
    assert (d <= tol).all(), f"{what}: max excess {(d - tol).max().item():.3e}"
    assert (d > 0).float().mean().item() < frac, f"{what}: {(d > 0).float().mean().item():.2%} of elements differ"


def test_layout_converters_are_exact(dev):
    x = rnd(1, 3, 24, 37)
    xb = blocked(x, dev)
    assert xb.shape == (3, 3, 37, 8) and xb.dtype == torch.bfloat16
    assert torch.equal(xb.cpu(), x.to(torch.bfloat16).reshape(3, 3, 8, 37).permute(0, 1, 3, 2).contiguous())
    assert torch.equal(SAMP.to_planar(xb).cpu(), rb(x))


CONV_CASES = [
    # N, Cin, Lin, Cout, K, stride, pad
    (3, 64, 500, 64, 3, 1, 1), (2, 64, 500, 128, 3, 2, 1), (2, 64, 500, 128, 1, 2, 0), (3, 128, 250, 256, 3, 2, 1),
    (3, 256, 125, 512, 3, 2, 1), (3, 256, 125, 512, 1, 2, 0), (5, 512, 63, 512, 3, 1, 1), (5, 512, 63, 128, 3, 1, 1),
    (7, 64, 37, 64, 3, 1, 1), (7, 64, 38, 128, 3, 2, 1), (9, 128, 21, 256, 1, 2, 0), (2, 16, 300, 64, 3, 1, 1), (1, 64, 1, 64, 3, 1, 1),
    (130, 64, 5, 64, 3, 1, 1),
    # the LDS-DMA ring kernel (3 taps, stride 1, Cout % 128 == 0): two-stage K loop, tiles streaming into the next one
    # (more position tiles than workgroup columns: 70 * 250 / 256 = 69 > 64), a single partial tile, taps crossing many
    # sample boundaries inside one tile (L = 5, 37), a wave whose 64 positions lie wholly beyond the last position
    (5, 64, 63, 128, 3, 1, 1), (70, 128, 250, 512, 3, 1, 1), (1, 256, 37, 128, 3, 1, 1), (300, 64, 5, 256, 3, 1, 1),
    (2, 128, 100, 128, 3, 1, 1), (1, 64, 1, 128, 3, 1, 1),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_b16_fwd_dgrad_wgrad(case, dev):
    N, Cin, Lin, Cout, K, s, p = case
    x = rb(rnd(1, N, Cin, Lin)).requires_grad_(True)
    w = rnd(2, Cout, Cin, K, std=(2.0 / (K * Cout)) ** 0.5)
    wr = rb(w).requires_grad_(True)
    y_ref = F.conv1d(x, wr, stride=s, padding=p)
    Lout = y_ref.shape[2]
    dy = rb(rnd(3, N, Cout, Lout))
    dx_ref, dw_ref = torch.autograd.grad(y_ref, (x, wr), dy)
    wg = w.to(dev)
    ops.begin_forward()
    yb, stats = SAMP.conv_fwd(blocked(x.detach(), dev), wg, s, p, want_stats=True)
    y = SAMP.to_planar(yb)
    assert_bf16_close(y, y_ref, "forward")
    sums = ops.bn_reduce_partials(stats).cpu()
    yd = y.double().cpu()
    ref_q = (yd ** 2).sum(dim=(0, 2))
    assert ((sums[:, 0] - yd.sum(dim=(0, 2))).abs().max() / (ref_q.sqrt().max() + 1e-30)).item() < 1e-4
    assert rel(sums[:, 1], ref_q) < 2e-5                              # statistics of the ROUNDED output
    if Cin % 64 == 0:
        dyb = blocked(dy, dev)
        dxb = SAMP.conv_dgrad(dyb, wg, Lin, s, p)
        assert_bf16_close(SAMP.to_planar(dxb), dx_ref, "data gradient")
        acc = rb(rnd(4, N, Cin, Lin))
        dx2 = SAMP.conv_dgrad(dyb, wg, Lin, s, p, accumulate=blocked(acc, dev))
        assert_bf16_close(SAMP.to_planar(dx2), dx_ref + acc, "data gradient + accumulate")
        dw = SAMP.conv_wgrad(dyb, blocked(x.detach(), dev), K, s, p)
        assert rel(dw, dw_ref) < 2e-5, "weight gradient (fp32 accumulation of exact bf16 products)"
        assert torch.equal(dw, SAMP.conv_wgrad(dyb, blocked(x.detach(), dev), K, s, p))   # fixed slab order: reproducible


@pytest.mark.parametrize("shape", [(4, 64, 500), (3, 256, 125), (5, 512, 63), (2, 8, 37)])
@pytest.mark.parametrize("relu,use_res", [(True, False), (True, True), (False, False)])
def test_bn_b16_fwd_bwd(shape, relu, use_res, dev):
    N, C, L = shape
    x = rb(rnd(1, N, C, L) * 1.7 + 0.4).requires_grad_(True)
    g = (1.0 + 0.2 * rnd(2, C)).requires_grad_(True); b = (0.1 * rnd(3, C)).requires_grad_(True)
    res = rb(rnd(4, N, C, L)).requires_grad_(True) if use_res else None
    z = F.batch_norm(x, None, None, g, b, training=True, momentum=0.1, eps=1e-5)
    if use_res: z = z + res
    y_ref = F.relu(z) if relu else z
    dy = rb(rnd(7, N, C, L))
    # the reference backward starts from the ROUNDED output's mask, like the kernel
    grads = torch.autograd.grad(y_ref, (x, g, b) + ((res,) if use_res else ()), dy)
    xd = x.detach().double()
    mean = xd.mean(dim=(0, 2)).float().to(dev)
    invstd = (xd.var(dim=(0, 2), unbiased=False) + 1e-5).rsqrt().float().to(dev)
    xb = blocked(x.detach(), dev)
    gg, bg = g.detach().to(dev), b.detach().to(dev)
    resb = blocked(res.detach(), dev) if use_res else None
    yb = SAMP.bn_apply_fwd(xb, mean, invstd, gg, bg, resb, relu)
    assert_bf16_close(SAMP.to_planar(yb), y_ref, "bn forward")
    dyb = blocked(dy, dev)
    mode = 0 if not relu else (1 if use_res else 2)
    part = SAMP.bn_bwd_reduce(dyb, yb if mode == 1 else None, xb, mean, invstd, gg, bg, mode)
    sums, dgam, dbet = ops.bn_reduce_partials(part, want_param_grads=True)
    assert rel(dgam, grads[1]) < 1e-3 and rel(dbet, grads[2]) < 1e-3        # (mask near-ties at |z| ~ 0 excepted)
    dx, dz = SAMP.bn_bwd_apply(dyb, yb if mode == 1 else None, xb, mean, invstd, gg, bg, mode, sums, N * L, want_dz=use_res)
    got, ref = SAMP.to_planar(dx).cpu(), grads[0]
    assert ((got - rb(ref)).abs().max() / ref.abs().max()).item() < 2e-2
    assert ((got - ref).norm() / ref.norm()).item() < 5e-3
    if use_res:
        assert_bf16_close(SAMP.to_planar(dz), grads[3], "residual gradient")


def _amp_model(C, sd_np, dev):
    return SAMP.enable(build_hip_model(C, sd_np, dev))


@pytest.mark.parametrize("C,B,seed", [(12, 2, 5), (1, 3, 6)])
def test_amp_fixmatch_step_against_emulation(C, B, seed, dev):
    """Whole FixMatch step with the student pass on the bf16 path vs oracle/amp_ref.py (same rounding points, CPU fp32
    arithmetic).  Teacher pass is fp32 in both (outside autocast) -> pseudo-labels as tight as the fp32 path."""
    from oracle import amp_ref as A
    from oracle import torch_ref as O
    L = 2000
    sd_np = synth.model_state(seed, C, trained=True, sharpen=1.0)
    batch_np = synth.fixmatch_batch(seed + 1, B, C, L)
    dm_np = dropout_mask_np(seed + 1, 2 * B)
    with torch.no_grad():
        conf0 = O.pseudo_label(O.model_forward(O.state_from_numpy(sd_np, requires_grad=False),
                                               torch.from_numpy(batch_np["unlabeled"]["ecg"]), train=False))[0]
    thr = round(float(conf0.median()), 3)
    cfg = dict(TRAIN_CFG, conf_thresh=thr); ocfg = dict(cfg, betas=(0.9, 0.999))
    o_sd = O.state_from_numpy(sd_np)
    r = A.fixmatch_step(o_sd, {}, cpu_batch(batch_np), ocfg, 3.0, torch.from_numpy(dm_np.astype(np.float32)))
    r32 = O.fixmatch_step(O.state_from_numpy(sd_np), {}, cpu_batch(batch_np), ocfg, 3.0, torch.from_numpy(dm_np.astype(np.float32)))
    model = _amp_model(C, sd_np, dev)
    model.decode_head.fixed_dropout_mask = torch.from_numpy(dm_np).to(dev, torch.uint8)
    b = to_dev(batch_np, dev)
    with torch.no_grad():
        model.eval()
        pred_u_w = model(b["unlabeled"]["ecg"], return_loss=False)["seg_logits"]
        conf, mask, _ = SF.pseudo_label(pred_u_w)
    assert rel(pred_u_w, r["pred_u_w"]) < 1e-4                      # fp32 teacher pass
    model.train()
    logits = model(torch.cat((b["labeled"]["ecg"], b["unlabeled"]["ecg_aug"])), return_loss=False)["seg_logits"]
    assert logits.dtype == torch.float32
    e_logits = rel(logits, r["logits"])
    e_vs_fp32 = rel(r["logits"], r32["logits"])
    e_hip_fp32 = rel(logits, r32["logits"])
    loss, stats = SF.fixmatch_loss(logits, B, b["labeled"]["target"], mask, conf, thr)
    st = stats.cpu().numpy()
    e_loss = max(abs(st[j] - r[k]) / max(abs(r[k]), 1e-3) for j, k in enumerate(("loss_total", "loss_x", "loss_u_s")))
    loss.backward()
    SF.wait_for_wgrads()

    def l2(a, b_):
        return ((a.double() - b_.double()).norm() / (b_.double().norm() + 1e-300)).item()

    g_hip = {k: p.grad.detach().cpu() for k, p in model.named_parameters()}
    hip_vs_emu = float(np.median([l2(g_hip[k], r["grads"][k]) for k in g_hip]))
    hip_vs_f32 = float(np.median([l2(g_hip[k], r32["grads"][k]) for k in g_hip]))
    emu_vs_f32 = float(np.median([l2(r["grads"][k], r32["grads"][k]) for k in g_hip]))
    print(f"amp C={C} B={B}: logits HIP-vs-emulation {e_logits:.2e}, HIP-vs-fp32 {e_hip_fp32:.2e}, emulation-vs-fp32 {e_vs_fp32:.2e}; "
          f"losses {e_loss:.2e}; gradient median rel. L2: HIP-vs-emulation {hip_vs_emu:.2e}, HIP-vs-fp32 {hip_vs_f32:.2e}, "
          f"emulation-vs-fp32 {emu_vs_f32:.2e}")
    # Rounding to bf16 is a chaotic map: two evaluations that differ by a relative delta before a rounding differ by
    # ~sqrt(2^-8 * delta) after it, so after the ~45 roundings of this network ANY two correct implementations of the same
    # policy (this kernel path, the CPU emulation) are as far from each other as each is from the fp32 result.  What can
    # be asserted at model level: the losses agree (1e-2 class; measured ~6e-4), and the HIP path deviates from the fp32
    # oracle no more than the emulation of the same policy does.  The unit-level test below pins single blocks tightly.
    assert e_loss < 1e-2
    assert e_hip_fp32 < 1.5 * e_vs_fp32 + 1e-3
    assert hip_vs_f32 < 1.5 * emu_vs_f32 + 1e-3
    sd = model.state_dict()
    for k, v in sd.items():
        if "running" in k:
            assert rel(v, o_sd[k]) < 2e-3, k                        # BN statistics of rounded tensors


@pytest.mark.parametrize("cin,cout,stride,L,N", [(64, 64, 1, 500, 3), (64, 128, 2, 500, 3), (256, 512, 2, 125, 4), (512, 512, 1, 63, 5)])
def test_amp_basic_block_against_emulation(cin, cout, stride, L, N, dev):
    """ONE BasicBlock (5-7 roundings) forward + backward from identical bf16 inputs vs oracle/amp_ref._basic_block: the
    outputs may differ only by isolated 1-ulp bf16 flips -> 1e-2 class bars on every element, 2e-3 on relative L2."""
    from collections import OrderedDict

    from models.backbones.resnet import BasicBlock
    from oracle import amp_ref as A
    from ssecg.nn import BatchNorm1d, Conv1d
    import torch.nn as nn
    torch.manual_seed(cin + cout + L)
    ds = None
    if stride != 1 or cin != cout:
        ds = nn.Sequential(Conv1d(cin, cout, 1, stride=stride, bias=False), BatchNorm1d(cout))
    blk = BasicBlock(cin, cout, stride, 1, ds)
    with torch.no_grad():
        for m in blk.modules():
            if isinstance(m, Conv1d):
                m.weight.normal_(0, (2.0 / (m.kernel_size[0] * m.out_channels)) ** 0.5)
            if isinstance(m, nn.BatchNorm1d):
                m.weight.copy_(1.0 + 0.2 * torch.randn_like(m.weight)); m.bias.copy_(0.1 * torch.randn_like(m.bias))
    sd = OrderedDict(("b." + k, v.detach().clone().requires_grad_(v.is_floating_point() and "running" not in k))
                     for k, v in blk.state_dict().items())
    x = rb(torch.relu(rnd(1, N, cin, L)))
    xr = x.clone().requires_grad_(True)
    out_ref = A._basic_block(sd, "b", xr, stride, ds is not None)
    dy = rb(rnd(2, *out_ref.shape))
    out_ref.backward(dy)
    blk = SAMP.enable(blk.to(dev)).train()
    ops.begin_forward()
    xb = SAMP.to_blocked(x.to(dev)).requires_grad_(True)
    out = blk(xb)
    out.backward(SAMP.to_blocked(dy.to(dev)))
    SF.flush_counters()

    def l2(a, b_):
        return ((a.detach().double().cpu() - b_.detach().double()).norm() / (b_.detach().double().norm() + 1e-300)).item()

    o = SAMP.to_planar(out.detach()).cpu()
    assert ((o - out_ref.detach()).abs().max() / out_ref.detach().abs().max()).item() < 2e-2
    assert l2(o, out_ref) < 2e-3, l2(o, out_ref)
    dxp = SAMP.to_planar(xb.grad).cpu()
    assert l2(dxp, xr.grad) < 5e-3, l2(dxp, xr.grad)
    for k, prm in blk.named_parameters():
        e = l2(prm.grad, sd["b." + k].grad)
        assert e < 1e-2, (k, e)
    for k, v in blk.state_dict().items():
        if "running" in k:
            assert rel(v, sd["b." + k]) < 1e-3, k


def test_amp_training_tracks_fp32(dev):
    """40 FixMatch + AdamW steps from the same init with the fp32 and the bf16 path: both loss curves settle together."""
    import algorithms.fixmatch as A_fm
    from utils.optimizer import get_optimizer_from_config
    C, B, L, seed, steps = 2, 16, 2000, 77, 40
    sd_np = synth.model_state(seed, C, trained=False)
    curves = {}
    for amp in (False, True):
        model = build_hip_model(C, sd_np, dev)
        if amp:
            SAMP.enable(model)
        model.decode_head.dropout = None; model.decode_head.dropout_ratio = 0.0
        opt = get_optimizer_from_config(dict(TRAIN_CFG, lr=1e-3), model.parameters())
        hist = []
        for s in range(steps):
            b = to_dev(synth.fixmatch_batch(seed + 1 + s, B, C, L), dev)
            loss, stats = A_fm.fixmatch_step(model, b["labeled"]["ecg"], b["labeled"]["target"], b["unlabeled"]["ecg"],
                                             b["unlabeled"]["ecg_aug"], TRAIN_CFG["conf_thresh"])
            loss.backward()
            SF.wait_for_wgrads()
            opt.step(); opt.zero_grad()
            hist.append(stats[:2].clone())
        curves[amp] = torch.stack(hist).cpu().numpy()
    a, b = curves[True], curves[False]
    assert np.isfinite(a).all() and np.isfinite(b).all()
    assert abs(a[0, 1] - b[0, 1]) < 2e-2 * b[0, 1]
    tail = slice(steps - 15, steps)
    print("loss_x first / last-15 mean: fp32", b[0, 1], b[tail, 1].mean(), " bf16", a[0, 1], a[tail, 1].mean())
    assert abs(a[tail, 1].mean() - b[tail, 1].mean()) < 0.05 * b[tail, 1].mean()
    assert a[tail, 1].mean() < a[:3, 1].mean()


def test_use_amp_flag_selects_the_bf16_path_in_the_plugins(dev):
    """``use_amp=True`` passed to the plugins' train_one_epoch (the reference's autocast switch) runs the student pass on the
    bf16 kernels and ``use_amp=False`` switches the same model back to fp32; losses agree at the 1e-2 class."""
    import algorithms.fixmatch as A_fm
    from utils.misc import NativeScalerWithGradNormCount
    from utils.optimizer import get_optimizer_from_config
    C, B, seed = 2, 2, 9
    sd_np = synth.model_state(seed, C, trained=True, sharpen=16.0)
    b = to_dev(synth.fixmatch_batch(seed + 1, B, C, 2000), dev)
    out = {}
    for amp in (True, False):
        model = build_hip_model(C, sd_np, dev)
        model.decode_head.fixed_dropout_mask = torch.from_numpy(dropout_mask_np(seed + 1, 2 * B)).to(dev, torch.uint8)
        opt = get_optimizer_from_config(dict(TRAIN_CFG), model.parameters())
        seen = []
        h = model.backbone.layer1[0].register_forward_hook(lambda m, i, o: seen.append((m.training, o.dtype)))
        out[amp] = A_fm.train_one_epoch(model, [b["labeled"]], [b["unlabeled"]], opt, dev, 3, NativeScalerWithGradNormCount(), None,
                                        amp, dict(TRAIN_CFG))
        h.remove()
        # teacher pass (eval) fp32, student pass bf16 iff use_amp
        assert seen == [(False, torch.float32), (True, torch.bfloat16 if amp else torch.float32)]
    for k in ("loss_total", "loss_x", "loss_u_s"):
        assert abs(out[True][k] - out[False][k]) < 2e-2 * max(abs(out[False][k]), 1e-3), (k, out[True][k], out[False][k])
    assert abs(out[True]["mask_ratio"] - out[False]["mask_ratio"]) < 1e-6       # pseudo-labels come from the fp32 teacher pass
