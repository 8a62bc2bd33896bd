"""GPU: size-independent properties at the BASELINE.json sizes (N = 1024 student windows per conv launch, B = 512-class
batches) where the CPU oracle is too slow: the two independent convolution implementations (direct implicit GEMM and the
Winograd kernels the network's shapes select - F(4,3) forward / data gradient since rounds 2-4, the F(2,3)-transpose weight
gradient) agree, convolution is linear in its input, the weight gradient is the adjoint of the forward, and a whole FixMatch
step gives the same losses and gradients with either implementation."""
import numpy as np
import pytest
import torch

from helpers import TRAIN_CFG, build_hip_model, sharpen_for, to_dev
from ssecg import functional as SF
from ssecg import ops, synth

pytestmark = pytest.mark.gpu

LAYERS = [(1024, 64, 500, 64), (1024, 128, 250, 128), (1024, 256, 125, 256), (1024, 512, 63, 512), (1024, 512, 63, 128)]


def _rel(a, b):
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


@pytest.mark.parametrize("N,C,L,M", LAYERS)
def test_winograd_and_direct_kernels_agree_at_full_size(N, C, L, M, dev, monkeypatch):
    g = torch.Generator(device="cpu").manual_seed(N + C + L)
    x = torch.randn(N, C, L, generator=g).to(dev)
    w = (torch.randn(M, C, 3, generator=g) * (2.0 / (3 * M)) ** 0.5).to(dev)
    dy = torch.randn(N, M, L, generator=g).to(dev)
    res = {}
    for mode in (True, False):
        monkeypatch.setattr(ops, "WINOGRAD", mode)
        y, stats = ops.conv1d_fwd(x, w, 1, 1, 1, want_stats=True)
        res[mode] = (y, ops.bn_reduce_partials(stats), ops.conv1d_dgrad(dy, w, L, 1, 1, 1), ops.conv1d_wgrad(dy, x, 3, 1, 1, 1))
    for a, b, what in zip(res[True], res[False], ("forward", "BN sums", "data gradient", "weight gradient")):
        assert _rel(a.double(), b.double()) < 2e-5, what
    # linearity in the input and adjointness <conv(x), dy> == <w, wgrad(dy, x)> == <x, dgrad(dy)>  (fp64 inner products)
    monkeypatch.setattr(ops, "WINOGRAD", True)
    y, _, dx, dw = res[True]
    x2 = torch.randn(N, C, L, generator=g).to(dev)
    y2, _ = ops.conv1d_fwd(x2, w, 1, 1, 1)
    y12, _ = ops.conv1d_fwd(0.5 * x + x2, w, 1, 1, 1)
    assert _rel(y12, 0.5 * y + y2) < 2e-5
    ip = (y.double() * dy.double()).sum().item()
    assert abs((w.double() * dw.double()).sum().item() - ip) < 1e-6 * (y.double().norm() * dy.double().norm()).item()
    assert abs((x.double() * dx.double()).sum().item() - ip) < 1e-6 * (y.double().norm() * dy.double().norm()).item()


def test_fixmatch_step_is_the_same_with_either_conv_algorithm(dev, monkeypatch):
    import algorithms.fixmatch as A_fm
    C, B, L, seed = 12, 64, 2000, 31
    sd_np = synth.model_state(seed, C, trained=True, sharpen=sharpen_for(C))
    batch = to_dev(synth.fixmatch_batch(seed + 1, B, C, L), dev)
    out = {}
    for mode in (True, False):
        monkeypatch.setattr(ops, "WINOGRAD", mode)
        model = build_hip_model(C, sd_np, dev)
        torch.manual_seed(0)
        model.decode_head.fixed_dropout_mask = (torch.rand(2 * B, 128, 63, generator=torch.Generator().manual_seed(3)) >= 0.1).to(dev, torch.uint8)
        loss, stats = A_fm.fixmatch_step(model, batch["labeled"]["ecg"], batch["labeled"]["target"], batch["unlabeled"]["ecg"],
                                         batch["unlabeled"]["ecg_aug"], TRAIN_CFG["conf_thresh"])
        loss.backward()
        out[mode] = (stats.cpu().numpy(), {k: p.grad.detach().clone() for k, p in model.named_parameters()})
    sa, sb = out[True][0], out[False][0]
    assert np.abs(sa - sb).max() < 1e-4 * max(np.abs(sb).max(), 1e-3), (sa, sb)
    worst = 0.0
    for k, ga in out[True][1].items():
        gb = out[False][1][k]
        l2 = ((ga - gb).double().norm() / (gb.double().norm() + 1e-30)).item()
        worst = max(worst, l2)
        assert l2 < 2e-2, f"{k}: relative L2 {l2:.2e}"      # ReLU near-ties may flip single channels (see helpers.check_packed)
    print(f"worst gradient relative L2 difference between the two conv algorithms: {worst:.2e}")


def test_training_trajectories_agree_between_conv_algorithms(dev, monkeypatch):
    """60 FixMatch + AdamW steps (B = 32, C = 2, fresh synthetic batch every step) with the direct and the Winograd
    convolution kernels from the same initial weights: the two loss curves must stay together (no drift / instability
    from the faster algorithm).  Individual steps are chaotic, so the comparison is on the curve, not bitwise."""
    import algorithms.fixmatch as A_fm
    from utils.optimizer import get_optimizer_from_config
    C, B, L, seed, steps = 2, 32, 2000, 77, 60
    sd_np = synth.model_state(seed, C, trained=False)
    curves = {}
    for mode in (True, False):
        monkeypatch.setattr(ops, "WINOGRAD", mode)
        model = build_hip_model(C, sd_np, dev)
        model.decode_head.dropout = None; model.decode_head.dropout_ratio = 0.0
        opt = get_optimizer_from_config(dict(TRAIN_CFG, lr=1e-3), model.parameters())
        for g in opt.param_groups:
            g["lr"] = 1e-3
        hist = []
        for s in range(steps):
            b = to_dev(synth.fixmatch_batch(seed + 1 + s, B, C, L), dev)
            loss, stats = A_fm.fixmatch_step(model, b["labeled"]["ecg"], b["labeled"]["target"], b["unlabeled"]["ecg"],
                                             b["unlabeled"]["ecg_aug"], TRAIN_CFG["conf_thresh"])
            loss.backward()
            opt.step(); opt.zero_grad()
            hist.append(stats[:2].clone())
        curves[mode] = torch.stack(hist).cpu().numpy()
    a, b = curves[True], curves[False]
    assert np.isfinite(a).all() and np.isfinite(b).all()
    assert np.abs(a[:3] - b[:3]).max() < 1e-3 * np.abs(b[:3]).max()                 # the first steps still coincide
    tail = slice(steps - 20, steps)
    assert abs(a[tail, 1].mean() - b[tail, 1].mean()) < 0.05 * b[tail, 1].mean(), (a[tail, 1].mean(), b[tail, 1].mean())
    # labels are independent of the signal in the synthetic data: the supervised loss settles at ln 4 from above
    assert b[tail, 1].mean() < b[:5, 1].mean() and abs(b[tail, 1].mean() - np.log(4.0)) < 0.05
    print("loss_x, first 5 / last 20 steps:", b[:5, 1].mean(), b[tail, 1].mean(), "(winograd:", a[tail, 1].mean(), ")")


def test_step_is_bitwise_reproducible(dev):
    """No atomics anywhere on the path: slab / partial sums are combined in fixed orders, so the same step from the same
    state gives bit-identical losses, gradients and BatchNorm buffers (B = 64, C = 12)."""
    import algorithms.fixmatch as A_fm
    C, B, L, seed = 12, 64, 2000, 41
    sd_np = synth.model_state(seed, C, trained=True, sharpen=sharpen_for(C))
    batch = to_dev(synth.fixmatch_batch(seed + 1, B, C, L), dev)
    runs = []
    for _ in range(2):
        model = build_hip_model(C, sd_np, dev)
        model.decode_head.fixed_dropout_mask = (torch.rand(2 * B, 128, 63, generator=torch.Generator().manual_seed(3)) >= 0.1).to(dev, torch.uint8)
        loss, stats = A_fm.fixmatch_step(model, batch["labeled"]["ecg"], batch["labeled"]["target"], batch["unlabeled"]["ecg"],
                                         batch["unlabeled"]["ecg_aug"], TRAIN_CFG["conf_thresh"])
        loss.backward()
        runs.append((stats.clone(), {k: p.grad.detach().clone() for k, p in model.named_parameters()},
                     {k: v.detach().clone() for k, v in model.state_dict().items() if "running" in k}))
    assert torch.equal(runs[0][0], runs[1][0])
    for k in runs[0][1]:
        assert torch.equal(runs[0][1][k], runs[1][1][k]), k
    for k in runs[0][2]:
        assert torch.equal(runs[0][2][k], runs[1][2][k]), k


def test_bench_size_forward_splits_into_oracle_checked_chunks(dev, monkeypatch):
    """B = 512 (the bench's per-GPU batch), C = 12, L = 2000.  Eval mode: every window is independent, so the logits of the
    full batch must equal, BIT FOR BIT, those of the same windows pushed through in chunks of 32 - the size at which
    tests/test_parity_r2_gpu.py pins the same path against the CPU oracle.  Train mode: BatchNorm couples the windows, so the
    512-batch statistics are checked against an fp64 accumulation of the chunked conv outputs instead (stem conv + BN).
    The K split of small launches (default on since round 5) changes the summation ORDER of a 32-window chunk's convolutions,
    not the kernels of the 512-batch: bit for bit with the split off (SSECG_KSPLIT=0), and with it on the chunks' logits sit
    within 2e-5 of the full batch's and every arg-max with a top-2 margin above 1e-4 is the same."""
    C, B, L, seed = 12, 512, 2000, 77
    sd_np = synth.model_state(seed, C, trained=True)
    model = build_hip_model(C, sd_np, dev).eval()
    x = torch.from_numpy(synth.fixmatch_batch(seed + 1, B, C, L)["unlabeled"]["ecg"]).to(dev)
    with torch.no_grad():
        full = model(x, return_loss=False)["seg_logits"]
        assert ops.KSPLIT
        split_parts = torch.cat([model(x[i:i + 32], return_loss=False)["seg_logits"] for i in range(0, B, 32)])
        monkeypatch.setattr(ops, "KSPLIT", False)
        assert torch.equal(full, model(x, return_loss=False)["seg_logits"])       # no launch of the full batch is split
        parts = torch.cat([model(x[i:i + 32], return_loss=False)["seg_logits"] for i in range(0, B, 32)])
    assert torch.isfinite(full).all()
    assert torch.equal(full, parts)
    assert not torch.equal(full, split_parts) and _rel(split_parts, full) < 2e-5
    # evaluate() / inference.py never split a launch (ops.ksplit_disabled): with the switch at its default a record's
    # logits through that path are bit-identical whatever batch it shares - validation metrics cannot flip near-tie arg-maxes with
    # the dataloader's batch size; under use_amp (the 16-bit eval path: no split exists there) the same holds
    monkeypatch.setattr(ops, "KSPLIT", True)
    import algorithms.base as A_base
    from ssecg import amp as SAMP
    cap = []
    h = model.register_forward_hook(lambda m, i, o: cap.append(o["seg_logits"].detach().clone()) or None)
    labels = torch.zeros((B, L), dtype=torch.int64)
    for use_amp in (False, True):
        outs = {}
        for bs in (32, 128):
            cap.clear()
            loader = [{"ecg": x[i:i + bs], "target": labels[i:i + bs]} for i in range(0, 256, bs)]
            A_base.evaluate(model, loader, dev, None, use_amp=use_amp, return_outputs=False)
            outs[bs] = torch.cat(cap)
        assert torch.equal(outs[32], outs[128]), f"evaluate(use_amp={use_amp}): a record's logits depend on the batch size"
        if not use_amp:
            assert torch.equal(outs[32], full[:256])
    h.remove()
    monkeypatch.setattr(ops, "KSPLIT", False)
    top2 = full.topk(2, dim=1)[0]
    clear = (top2[:, 0] - top2[:, 1]) > 1e-4
    assert clear.float().mean().item() > 0.99 and torch.equal(split_parts.argmax(dim=1)[clear], full.argmax(dim=1)[clear])
    # pseudo-label head at full size: argmax / confidence of the full batch == of the chunks
    c1, m1, _ = SF.pseudo_label(full)
    c2, m2, _ = SF.pseudo_label(parts)
    assert torch.equal(m1, m2) and torch.equal(c1, c2)
    # train-mode stem statistics of the 512-batch vs fp64 sums over chunked conv outputs
    w = model.backbone.stem[0].weight.detach()
    y, partial = ops.conv1d_fwd(x, w, 2, 3, 1, want_stats=True)
    s = ops.bn_reduce_partials(partial).double().view(-1, 2)
    ref_s = torch.zeros(64, dtype=torch.float64, device=dev); ref_q = torch.zeros_like(ref_s)
    for i in range(0, B, 64):
        yc, _ = ops.conv1d_fwd(x[i:i + 64], w, 2, 3, 1)
        assert torch.equal(yc, y[i:i + 64])
        ref_s += yc.double().sum((0, 2)); ref_q += (yc.double() ** 2).sum((0, 2))
    assert torch.allclose(s[:, 0], ref_s, rtol=1e-5, atol=1e-3) and torch.allclose(s[:, 1], ref_q, rtol=1e-5)


@pytest.mark.parametrize("C,L,M", [(128, 250, 128), (512, 63, 512)])
def test_bf16_ring_kernels_at_full_size_split_into_checked_chunks(C, L, M, dev):
    """N = 1024 (the bench's student batch): the LDS-DMA ring kernels of the bf16 path tile the FLATTENED position axis, so a
    window's outputs must not depend on which other windows share its launch - forward (incl. BN partial sums) and data gradient
    of the full batch equal, bit for bit, those of 64-window chunks, the size class tests/test_amp_gpu.py checks against torch;
    the weight gradient (fp32 slab sums, different slab partition) agrees to 1e-5."""
    from ssecg import amp as SAMP
    N = 1024
    g = torch.Generator(device="cpu").manual_seed(C + L)
    x = SAMP.to_blocked(torch.randn(N, C, L, generator=g).to(dev))
    dy = SAMP.to_blocked(torch.randn(N, M, L, generator=g).to(dev))
    w = (torch.randn(M, C, 3, generator=g) * (2.0 / (3 * M)) ** 0.5).to(dev)
    ops.begin_forward()
    y, stats = SAMP.conv_fwd(x, w, 1, 1, want_stats=True)
    dx = SAMP.conv_dgrad(dy, w, L, 1, 1)
    dw = SAMP.conv_wgrad(dy, x, 3, 1, 1)
    sums = ops.bn_reduce_partials(stats).double()
    ref_s = torch.zeros_like(sums)
    dw_ref = torch.zeros_like(dw, dtype=torch.float64)
    for i in range(0, N, 64):
        yc, sc = SAMP.conv_fwd(x[i:i + 64].contiguous(), w, 1, 1, want_stats=True)
        assert torch.equal(yc, y[i:i + 64])
        assert torch.equal(SAMP.conv_dgrad(dy[i:i + 64].contiguous(), w, L, 1, 1), dx[i:i + 64])
        ref_s += ops.bn_reduce_partials(sc).double()
        dw_ref += SAMP.conv_wgrad(dy[i:i + 64].contiguous(), x[i:i + 64].contiguous(), 3, 1, 1).double()
    assert torch.allclose(sums, ref_s, rtol=1e-5, atol=1e-2)
    assert _rel(dw.double(), dw_ref) < 1e-5
