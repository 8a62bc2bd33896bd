"""GPU: the reduced-precision path (``use_amp: true``): bf16 blocked storage + bf16 MFMA kernels of
csrc/amp.hip against torch fp32 arithmetic on bf16-rounded operands (kernel level), and the whole student step against the
CPU emulation ``oracle/amp_ref.py`` (model level).  The reference's CUDA autocast cannot run without CUDA and the reference
holds no fixtures for it: PARITY UNPINNED for this row; the bars below are against the emulation."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import TRAIN_CFG, build_hip_model, cpu_batch, dropout_mask_np, learnable_batch as _learnable_batch, rel, to_dev
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


def assert_bf16_close(got, ref, what, frac=0.02, scale=None):
    """``got`` (fp32 view of a bf16 result) vs the fp32 reference rounded to bf16: identical up to 1 bf16 ulp on the few
    elements whose fp32 value sits at a rounding boundary (different summation order).  ``scale``: magnitude whose ulp bounds the
    deviation where the result is a sum of ROUNDED terms (a 1-ulp flip of the larger term survives a cancelling sum)."""
    got, ref = got.detach().float().cpu(), rb(ref.detach().float().cpu())
    d = (got - ref).abs()
    mag = ref.abs() if scale is None else ref.abs() + scale.detach().float().cpu().abs()
    tol = mag * 2.0 ** -7 + 1e-30 + ref.abs().max() * 1e-6
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
    # the weights-stationary kernel (csrc/amp_ws.hip; 3 taps, stride 1, 64 / 128 / 256 source channels): workgroups that loop over
    # several position tiles (more tiles than position lanes), two and three channel groups per position tile, the 2 x 2 wave
    # layout (64-channel multiples that are not multiples of 128), eight stages per tile (256 -> 64)
    (300, 128, 250, 128, 3, 1, 1), (40, 256, 125, 256, 3, 1, 1), (33, 128, 250, 64, 3, 1, 1), (9, 256, 63, 192, 3, 1, 1),
    (150, 64, 500, 64, 3, 1, 1), (3, 256, 2, 512, 3, 1, 1),
    # ... its stride-2 / 1x1 instances (round 4): the first conv of a stage and the downsample branch (forward), the downsample's
    # data gradient (strided output), several tiles per workgroup, odd source lengths (125 -> 63: the source window SHRINKS at a
    # sample boundary), many samples per tile
    (40, 256, 125, 512, 3, 2, 1), (33, 64, 500, 128, 1, 2, 0), (60, 128, 21, 256, 1, 2, 0), (25, 128, 250, 256, 3, 2, 1),
    (300, 64, 7, 128, 3, 2, 1), (50, 256, 125, 512, 1, 2, 0),
]


def _random_b16_shapes(n=16, seed=777):
    """Seeded sweep: channel counts in multiples of 64 (what the ResNet body uses), odd lengths, 1 / 3 taps, stride 1 / 2,
    batches that leave partial position tiles or put many samples into one tile."""
    import numpy as np
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        k = int(rng.choice([1, 3, 3]))
        st = int(rng.choice([1, 2])) if k == 3 else 2
        out.append((int(rng.integers(1, 40)), 64 * int(rng.integers(1, 9)), int(rng.integers(2, 260)), 64 * int(rng.integers(1, 9)),
                    k, st, k // 2))
    return out


@pytest.mark.parametrize("case", CONV_CASES + _random_b16_shapes())
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
        # round 5: the branch's own gradient is rounded BEFORE the stored one is added (autograd adds two stored bf16 tensors)
        assert_bf16_close(SAMP.to_planar(dx2), rb(dx_ref) + acc, "data gradient + accumulate", scale=dx_ref)
        dw = SAMP.conv_wgrad(dyb, blocked(x.detach(), dev), K, s, p)
        assert rel(dw, dw_ref) < 2e-5, "weight gradient (fp32 accumulation of exact bf16 products)"
        assert torch.equal(dw, SAMP.conv_wgrad(dyb, blocked(x.detach(), dev), K, s, p))   # fixed slab order: reproducible


@pytest.mark.parametrize("force", ["0", "1"], ids=["kernels_of_amp_hip_everywhere", "weights_stationary_everywhere"])
@pytest.mark.parametrize("case", [(150, 64, 500, 64, 3, 1), (300, 128, 250, 128, 3, 1), (40, 256, 125, 256, 3, 1), (33, 128, 250, 64, 3, 1),
                                  (9, 256, 63, 192, 3, 1), (40, 64, 500, 128, 3, 2), (40, 256, 125, 512, 1, 2), (40, 128, 250, 256, 1, 2),
                                  # the 3-tap stride-2 data gradient's odd-position phase (2 taps) on the weights-stationary kernel, all three instances
                                  (40, 128, 250, 256, 3, 2), (24, 256, 125, 512, 3, 2)])
def test_conv_b16_both_kernel_families_on_the_shapes_they_share(case, force, dev, monkeypatch):
    """By default the weights-stationary kernel (csrc/amp_ws.hip) takes every forward it has an instance for but the plain 3-tap
    stride-1 data gradient only at 256 source channels (where it measured faster); SSECG_AMP_WS = 1 / 0 puts EVERY launch it can
    take on it / on the kernels of csrc/amp.hip, so both families are held to the same 1-ulp bar on the same shapes."""
    monkeypatch.setenv("SSECG_AMP_WS", force)
    N, Cin, Lin, Cout, K, st = case
    pad = K // 2
    x = rb(rnd(11, N, Cin, Lin)).requires_grad_(True)
    w = rnd(12, Cout, Cin, K, std=(2.0 / (K * Cout)) ** 0.5)
    wr = rb(w).requires_grad_(True)
    y_ref = F.conv1d(x, wr, stride=st, padding=pad)
    dy = rb(rnd(13, *y_ref.shape))
    dx_ref, = torch.autograd.grad(y_ref, (x,), dy)
    wg = w.to(dev)
    ops.begin_forward()
    yb, stats = SAMP.conv_fwd(blocked(x.detach(), dev), wg, st, pad, want_stats=True)
    y = SAMP.to_planar(yb)
    assert_bf16_close(y, y_ref, "forward")
    sums = ops.bn_reduce_partials(stats).cpu()
    yd = y.double().cpu()
    ref_q = (yd ** 2).sum(dim=(0, 2))
    assert ((sums[:, 0] - yd.sum(dim=(0, 2))).abs().max() / (ref_q.sqrt().max() + 1e-30)).item() < 1e-4
    assert rel(sums[:, 1], ref_q) < 2e-5
    dxb = SAMP.conv_dgrad(blocked(dy, dev), wg, Lin, st, pad)
    assert_bf16_close(SAMP.to_planar(dxb), dx_ref, "data gradient")


@pytest.mark.parametrize("case", [(96, 64, 500, 64, 3, 1), (96, 128, 250, 128, 3, 1), (96, 256, 125, 256, 3, 1), (96, 256, 125, 512, 3, 2),
                                  (96, 128, 250, 256, 1, 2), (128, 64, 63, 128, 3, 1)])
def test_ws_kernel_reproduces_bit_for_bit_under_memory_load(case, dev, monkeypatch):
    """Race screen for the hand-counted ``s_waitcnt vmcnt(N)`` / LDS-DMA ring of csrc/amp_ws.hip (cdna_hip_programming.md: "an
    early read passes reference checks whenever the DMA happens to land first - place reads by the count, never by clean runs",
    and: screen a new synchronisation structure over many runs at several sizes).  The kernel has no atomics and a fixed
    summation order, so ANY launch-to-launch difference is a race: 40 launches per shape (forward with statistics + data
    gradient, several tiles per workgroup) while a second stream streams 256 MB copies through HBM to perturb DMA arrival
    times; every output and every statistics row must equal the first launch's bit for bit, and the first launch is held to
    the usual 1-ulp bar against torch."""
    monkeypatch.setenv("SSECG_AMP_WS", "1")
    N, Cin, Lin, Cout, K, st = case
    pad = K // 2
    x = rb(rnd(21, N, Cin, Lin))
    w = rnd(22, Cout, Cin, K, std=(2.0 / (K * Cout)) ** 0.5)
    y_ref = F.conv1d(x, rb(w), stride=st, padding=pad)
    dy = rb(rnd(23, *y_ref.shape))
    xb, dyb, wg = blocked(x, dev), blocked(dy, dev), w.to(dev)
    ops.begin_forward()
    y0, s0 = SAMP.conv_fwd(xb, wg, st, pad, want_stats=True)
    assert_bf16_close(SAMP.to_planar(y0), y_ref, "forward")
    d0 = SAMP.conv_dgrad(dyb, wg, Lin, st, pad) if st == 1 or K == 1 else None
    y0, s0 = y0.clone(), s0.clone()
    d0 = d0.clone() if d0 is not None else None
    side = torch.cuda.Stream(device=dev)
    a, b = torch.empty(1 << 26, device=dev), torch.empty(1 << 26, device=dev)
    stop = 0
    for it in range(40):
        with torch.cuda.stream(side):
            for _ in range(1 + it % 3):
                b.copy_(a)                     # HBM traffic beside the kernel, a different amount every iteration
        y, s_ = SAMP.conv_fwd(xb, wg, st, pad, want_stats=True)
        assert torch.equal(y, y0) and torch.equal(s_, s0), f"forward launch {it} differs from launch 0"
        if d0 is not None:
            assert torch.equal(SAMP.conv_dgrad(dyb, wg, Lin, st, pad), d0), f"data-gradient launch {it} differs from launch 0"
        stop += 1
    torch.cuda.synchronize()
    assert stop == 40


@pytest.mark.parametrize("shape", [(4, 64, 500), (3, 256, 125), (5, 512, 63), (2, 8, 37)])
@pytest.mark.parametrize("relu,use_res", [(True, False), (True, True), (False, False)])
def test_bn_b16_fwd_bwd(shape, relu, use_res, dev):
    N, C, L = shape
    x = rb(rnd(1, N, C, L) * 1.7 + 0.4).requires_grad_(True)
    g = (1.0 + 0.2 * rnd(2, C)).requires_grad_(True); b = (0.1 * rnd(3, C)).requires_grad_(True)
    res = rb(rnd(4, N, C, L)).requires_grad_(True) if use_res else None
    z = F.batch_norm(x, None, None, g, b, training=True, momentum=0.1, eps=1e-5)
    if use_res: z = z + (rb(z) - z).detach() + res     # round 5: BatchNorm's output is rounded before ``out += identity`` (autocast's placement)
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
    if mode == 1:
        # the byte-per-vector ReLU mask of the apply pass (mode 3) drives both backward passes to bit-identical results
        yb2, mask = SAMP.bn_apply_fwd(xb, mean, invstd, gg, bg, resb, relu, want_mask=True)
        assert torch.equal(yb2, yb)
        want = yb.float() > 0                                                                      # blocked (N, C/8, L, 8)
        got = ((mask[..., None].to(torch.int32) >> torch.arange(8, device=mask.device)) & 1).bool()
        assert torch.equal(got, want)
        assert torch.equal(SAMP.bn_bwd_reduce(dyb, mask, xb, mean, invstd, gg, bg, 3), part)
        dx3, dz3 = SAMP.bn_bwd_apply(dyb, mask, xb, mean, invstd, gg, bg, 3, sums, N * L, want_dz=use_res)
        assert torch.equal(dx3, dx) and torch.equal(dz3, dz)
    got, ref = SAMP.to_planar(dx).cpu(), grads[0]
    assert ((got - rb(ref)).abs().max() / ref.abs().max()).item() < 2e-2
    assert ((got - ref).norm() / ref.norm()).item() < 5e-3
    if use_res:
        assert_bf16_close(SAMP.to_planar(dz), grads[3], "residual gradient")


def _amp_model(C, sd_np, dev):
    return SAMP.enable(build_hip_model(C, sd_np, dev))


def _l2(a, b_):
    a, b_ = a.detach().double().cpu(), b_.detach().double().cpu()
    return ((a - b_).norm() / (b_.norm() + 1e-300)).item()


def _cos(a, b_):
    a, b_ = a.detach().double().cpu().reshape(-1), b_.detach().double().cpu().reshape(-1)
    return (a @ b_ / (a.norm() * b_.norm() + 1e-300)).item()


@pytest.mark.parametrize("C,B,seed", [(12, 2, 5), (1, 3, 6)])
def test_amp_fixmatch_step_losses_and_teacher_pass(C, B, seed, dev):
    """Whole FixMatch step with the student pass on the bf16 path at the golden-fixture batch sizes: what is well
    conditioned there - the fp32 teacher pass (1e-4 vs the fp32 oracle: it is outside autocast), the loss values (1e-2 class
    vs the emulation), BN running statistics of the rounded tensors.  Gradients at B = 2-3 windows are dominated by bf16
    rounding noise for ANY implementation of the policy (relative L2 ~0.3 between the emulation and fp32), so they are judged
    where they can be: per unit / stage / head / stem boundary against the emulation (tests below, 1e-2 class) and as
    cosines against fp32 at B = 32 (test_amp_gradient_cosines_b32)."""
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
    loss, stats = SF.fixmatch_loss(logits, B, b["labeled"]["target"], mask, conf, thr)
    st = stats.cpu().numpy()
    e_loss = max(abs(st[j] - r[k]) / max(abs(r[k]), 1e-3) for j, k in enumerate(("loss_total", "loss_x", "loss_u_s")))
    print(f"amp C={C} B={B}: logits HIP-vs-emulation rel. L2 {_l2(logits, r['logits']):.2e}; losses {e_loss:.2e}")
    assert e_loss < 1e-2
    assert _l2(logits, r["logits"]) < 6e-2       # model depth: ~45 roundings amplify 1-ulp differences (see the unit tests)
    loss.backward()
    assert all(torch.isfinite(p.grad).all() for p in model.parameters())
    sd = model.state_dict()
    for k, v in sd.items():
        if "running" in k:
            assert rel(v, o_sd[k]) < 2e-3, k                        # BN statistics of rounded tensors


def _emu_state(sd_np, prefix):
    """oracle state restricted to ``prefix`` (tensors that need a gradient are leaves)."""
    from oracle import torch_ref as O
    return O.state_from_numpy({k: v for k, v in sd_np.items() if k.startswith(prefix)})


STAGES = [(1, 64, 500, 3), (2, 64, 500, 3), (3, 128, 250, 4), (4, 256, 125, 5)]
# Multi-unit bars: within 2.5 x the measured distance between two correct CPU evaluations of the policy (+1e-3), or inside the
# absolute caps below, whichever is larger.  The caps are 1.5 x the largest deviation measured on MI355X for a path whose
# every kernel is individually exact to <= 0.01 % of its outputs (profiles/r03_amp_flip_probe.txt): forward 1.35e-3 (stage 3),
# gradients 2.1e-2 (stem boundary); a 20 % error in any one dgrad / wgrad / BN-backward kernel gives >= 1e-1.
OUT_CAP, GRAD_CAP = 2e-3, 3e-2


def _emu_stage(sd_np, li, x, dy, acc=torch.float32):
    """-> (out, dx, {param: grad}, sd) of oracle/amp_ref over stage ``li`` with the convolutions accumulated in ``acc``."""
    from oracle import amp_ref as A
    pfx = f"backbone.layer{li}"
    sd = _emu_state(sd_np, pfx)
    xr = x.clone().requires_grad_(True)
    A.CONV_ACC, A.STAT_MODE = acc, ("exact" if acc is torch.float32 else "fp32_sequential")
    try:
        h = A._basic_block(sd, pfx + ".0", xr, 1 if li == 1 else 2, li > 1)
        out = A._basic_block(sd, pfx + ".1", h, 1, False)
        out.backward(dy)
    finally:
        A.CONV_ACC, A.STAT_MODE = torch.float32, "exact"
    return out.detach(), xr.grad, {k[len(pfx) + 1:]: v.grad for k, v in sd.items() if v.requires_grad}, sd


@pytest.mark.parametrize("li,cin,L,N", STAGES)
def test_amp_stage_against_emulation(li, cin, L, N, dev):
    """A whole ResNet STAGE (two BasicBlocks = 4-5 conv units, 10-14 roundings; stages 2-4 with the stride-2 conv, the 1x1
    downsample branch and its separately rounded input gradient) forward + backward from identical bf16 inputs, product
    modules vs oracle/amp_ref._basic_block.  Every kernel alone matches torch on the same bf16 operands to <= 0.01 % of
    its outputs (tools/amp_flip_probe.py, amp_block_probe.py; profiles/r03_amp_flip_probe.txt), but a 1-ulp bf16 flip
    changes ~400 sums of the next conv by 2e-4 relative and flips ~5 % of THEIR roundings: differences grow ~20x per unit.
    So the yardstick is measured in the test: a SECOND correct evaluation of the same policy (the emulation with fp64 conv
    accumulation and sequential-fp32 BN statistics - both sum in a different order, as any kernel does) sits at a distance
    ``floor`` from the first; the HIP path must be no further from the emulation than 2.5 x that floor (+1e-3), tensor by
    tensor, and inside absolute caps.  A wrong rounding point, a
    dropped residual gradient or a few-% dgrad/wgrad error in any one of the stage's kernels exceeds these."""
    C, seed = 2, 40 + li
    sd_np = synth.model_state(seed, C, trained=True, sharpen=1.0)
    model = _amp_model(C, sd_np, dev).train()
    stage = getattr(model.backbone, f"layer{li}")
    pfx = f"backbone.layer{li}"
    x = rb(torch.relu(rnd(li, N, cin, L)))
    dy = rb(rnd(li + 10, N, 64 * 2 ** (li - 1), L if li == 1 else (L + 1) // 2))
    out_ref, dx_ref, g_ref, sd = _emu_stage(sd_np, li, x, dy)
    out_alt, dx_alt, g_alt, _ = _emu_stage(sd_np, li, x, dy, acc=torch.float64)
    ops.begin_forward()
    xb = SAMP.to_blocked(x.to(dev)).requires_grad_(True)
    out = stage(xb)
    assert SAMP.is_blocked(out)
    out.backward(SAMP.to_blocked(dy.to(dev)))
    SF.flush_counters()
    o = SAMP.to_planar(out.detach()).cpu()
    e_out, f_out = _l2(o, out_ref), _l2(out_alt, out_ref)
    e_dx, f_dx = _l2(SAMP.to_planar(xb.grad), dx_ref), _l2(dx_alt, dx_ref)
    rows = [(k, _l2(prm.grad, g_ref[k]), _l2(g_alt[k], g_ref[k])) for k, prm in stage.named_parameters()]
    wk, worst, wfloor = max(rows, key=lambda t: t[1] / (2.5 * t[2] + 1e-3))
    print(f"stage {li}: output rel. L2 {e_out:.2e} (floor {f_out:.2e}), input gradient {e_dx:.2e} (floor {f_dx:.2e}), "
          f"worst parameter gradient {worst:.2e} (floor {wfloor:.2e}, {wk})")
    assert ((o - out_ref).abs().max() / out_ref.abs().max()).item() < 3e-2
    assert e_out < max(2.5 * f_out + 1e-3, OUT_CAP), (e_out, f_out)
    assert e_dx < max(2.5 * f_dx + 1e-3, GRAD_CAP), (e_dx, f_dx)
    for k, e, f in rows:
        assert e < max(2.5 * f + 1e-3, GRAD_CAP), (k, e, f)
    for k, v in stage.state_dict().items():
        if "running" in k:
            assert rel(v, sd[f"{pfx}.{k}"]) < 1e-3, k


def test_amp_head_unit_against_emulation(dev):
    """The FCN head under use_amp: bf16 conv unit (512 -> 128) on the blocked stage-4 feature map, then fp32 dropout (given
    mask) + 1x1 classifier, forward + backward vs the emulation."""
    import torch.nn.functional as F_
    from oracle import amp_ref as A
    C, N, Lf, seed = 2, 6, 63, 51
    sd_np = synth.model_state(seed, C, trained=True, sharpen=1.0)
    model = _amp_model(C, sd_np, dev).train()
    head = model.decode_head
    sd = _emu_state(sd_np, "decode_head")
    x = rb(torch.relu(rnd(1, N, 512, Lf)))
    xr = x.clone().requires_grad_(True)
    dm = dropout_mask_np(seed, N, lp=Lf)
    a = A._unit(sd, "decode_head.convs.0.0", "decode_head.convs.0.1", xr, 1, 1)
    a = a * torch.from_numpy(dm.astype(np.float32)) * (1.0 / 0.9)
    lo_ref = F_.conv1d(a, sd["decode_head.cls_seg.weight"], sd["decode_head.cls_seg.bias"])
    dy = rnd(2, *lo_ref.shape)
    lo_ref.backward(dy)
    head.fixed_dropout_mask = torch.from_numpy(dm).to(dev, torch.uint8)
    ops.begin_forward()
    xb = SAMP.to_blocked(x.to(dev)).requires_grad_(True)
    lo = head((None, None, None, xb))
    assert lo.dtype == torch.float32 and lo.shape == lo_ref.shape
    lo.backward(dy.to(dev))
    e_out, e_dx = _l2(lo, lo_ref), _l2(SAMP.to_planar(xb.grad), xr.grad)
    errs = {k: _l2(prm.grad, sd["decode_head." + k].grad) for k, prm in head.named_parameters()}
    print(f"head: logits rel. L2 {e_out:.2e}, input gradient {e_dx:.2e}, parameter gradients {max(errs.values()):.2e}")
    assert e_out < 3e-3 and e_dx < 5e-3
    assert max(errs.values()) < 1e-2, errs


def _emu_stem_stage1(sd_np, x, dy, acc=torch.float32):
    import torch.nn.functional as F_
    from oracle import amp_ref as A
    from oracle import torch_ref as R
    sd = _emu_state(sd_np, "backbone.")
    A.CONV_ACC, A.STAT_MODE = acc, ("exact" if acc is torch.float32 else "fp32_sequential")
    try:
        # the 16-bit stem of round 5 (bf16-rounded x and w, rounded conv output, rounded BN + ReLU + max-pool), summed in the other
        # order too: oracle/amp_ref.stem_forward under policy "hip" with this accumulation mode
        h = A.rb(A._conv(x.to(torch.bfloat16).to(torch.float32), A._w(sd["backbone.stem.0.weight"]), 2, 3))
        h = F_.relu(A.rb(A._bn_train(sd, "backbone.stem.1", h)))
        h = A._RoundGrad.apply(F_.max_pool1d(h, kernel_size=3, stride=2, padding=1))
        h = A._basic_block(sd, "backbone.layer1.0", h, 1, False)
        ref = A._basic_block(sd, "backbone.layer1.1", h, 1, False)
        ref.backward(dy)
    finally:
        A.CONV_ACC, A.STAT_MODE = torch.float32, "exact"
    return ref.detach(), {k[len("backbone."):]: v.grad for k, v in sd.items() if v.grad is not None}


def test_amp_stem_boundary_against_emulation(dev):
    """The stem on 16-bit operands (fp32 MFMA on bf16-rounded x and w, rounded output, fp32 statistics; round 5) -> blocked bf16
    layout -> stage 1, and back: the gradient crosses the boundary as exact fp32 of the stored bf16 values and reaches the stem
    weights / BN parameters (weight gradient from bf16-rounded x and dc).
    Bars as in the stage test: within 2.5 x the distance between two correct evaluations of the policy (+1e-3)."""
    C, N, L, seed = 12, 3, 2000, 61
    sd_np = synth.model_state(seed, C, trained=True, sharpen=1.0)
    model = _amp_model(C, sd_np, dev).train()
    x = rnd(3, N, C, L)
    dy = rb(rnd(4, N, 64, 500))
    ref, g_ref = _emu_stem_stage1(sd_np, x, dy)
    alt, g_alt = _emu_stem_stage1(sd_np, x, dy, acc=torch.float64)
    outs = model.backbone(x.to(dev))
    assert SAMP.is_blocked(outs[0])
    outs[0].backward(SAMP.to_blocked(dy.to(dev)))
    e_out, f_out = _l2(SAMP.to_planar(outs[0].detach()), ref), _l2(alt, ref)
    rows = [(k, _l2(prm.grad, g_ref[k]), _l2(g_alt[k], g_ref[k])) for k, prm in model.backbone.named_parameters()
            if k.startswith("stem") or k.startswith("layer1")]
    errs = {k: (e, f) for k, e, f in rows}
    print(f"stem boundary: stage-1 output rel. L2 {e_out:.2e} (floor {f_out:.2e}); stem weight gradient {errs['stem.0.weight'][0]:.2e} "
          f"(floor {errs['stem.0.weight'][1]:.2e}), worst of stem + stage 1 {max(e for _, e, _ in rows):.2e}")
    assert e_out < max(2.5 * f_out + 1e-3, OUT_CAP), (e_out, f_out)
    for k, e, f in rows:
        assert e < max(2.5 * f + 1e-3, GRAD_CAP), (k, e, f)


def test_amp_gradient_cosines_b32(dev):
    """FixMatch step at B = 32 labelled + 32 unlabelled windows, 12 leads, L = 2000 on the LEARNABLE synthetic task (labels
    are a function of the signal, so the batch gradient carries a coherent signal instead of label noise): per-tensor COSINE
    between the bf16 path's gradients and the fp32 oracle's.  The emulation of the same policy sets the expectation: the HIP
    path must be as well aligned with fp32 as the emulation is (-0.04, see below) on every tensor, and HIP and emulation must agree with
    each other better than either does with fp32 (>= 0.93; measured 0.96-0.999).  A dgrad / wgrad
    kernel with a 20 % error, a missing residual-branch gradient or a wrong BN backward lowers a cosine far below that."""
    from oracle import amp_ref as A
    from oracle import torch_ref as O
    C, B, L, seed = 12, 32, 2000, 93
    sd_np = synth.model_state(seed, C, trained=True, sharpen=1.0)
    batch_np = {k: v for k, v in _learnable_batch(seed + 1, B, C, L).items() if k != "u_target"}
    dm_np = dropout_mask_np(seed + 1, 2 * B)
    with torch.no_grad():
        conf0 = O.pseudo_label(O.model_forward(O.state_from_numpy(sd_np, requires_grad=False),
                                               torch.from_numpy(batch_np["unlabeled"]["ecg"]), train=False))[0]
    thr = round(float(conf0.median()), 3)
    cfg = dict(TRAIN_CFG, conf_thresh=thr); ocfg = dict(cfg, betas=(0.9, 0.999))
    dm = torch.from_numpy(dm_np.astype(np.float32))
    r32 = O.fixmatch_step(O.state_from_numpy(sd_np), {}, cpu_batch(batch_np), ocfg, 3.0, dm)
    remu = A.fixmatch_step(O.state_from_numpy(sd_np), {}, cpu_batch(batch_np), ocfg, 3.0, dm)
    model = _amp_model(C, sd_np, dev)
    model.decode_head.fixed_dropout_mask = torch.from_numpy(dm_np).to(dev, torch.uint8)
    b = to_dev(batch_np, dev)
    with torch.no_grad():
        model.eval()
        conf, mask, _ = SF.pseudo_label(model(b["unlabeled"]["ecg"], return_loss=False)["seg_logits"])
    model.train()
    logits = model(torch.cat((b["labeled"]["ecg"], b["unlabeled"]["ecg_aug"])), return_loss=False)["seg_logits"]
    loss, stats = SF.fixmatch_loss(logits, B, b["labeled"]["target"], mask, conf, thr)
    loss.backward()
    st = stats.cpu().numpy()
    for j, k in enumerate(("loss_total", "loss_x", "loss_u_s")):
        assert abs(st[j] - remu[k]) < 5e-3 * max(abs(remu[k]), 1e-3), (k, st[j], remu[k])
    rows = []
    for k, p in model.named_parameters():
        rows.append((k, _cos(p.grad, r32["grads"][k]), _cos(remu["grads"][k], r32["grads"][k]), _cos(p.grad, remu["grads"][k])))
    print("cosine vs fp32 (HIP, emulation) and HIP-vs-emulation, eight lowest:")
    for k, ch, ce, chh in sorted(rows, key=lambda t: t[1])[:8]:
        print(f"  {k:45s} {ch:.4f} {ce:.4f} {chh:.4f}")
    convs = [t for t in rows if t[0].endswith("conv1.weight") or t[0].endswith("conv2.weight") or t[0].endswith(".0.weight")]
    print(f"conv weights: lowest cosine to fp32 {min(t[1] for t in convs):.4f} (emulation {min(t[2] for t in convs):.4f})")
    print(f"HIP-vs-emulation: lowest cosine {min(t[3] for t in rows):.4f} (conv weights {min(t[3] for t in convs):.4f})")
    # measured on MI355X (B = 32, this seed): bf16 storage of activations AND gradients leaves the emulation itself at
    # cosine 0.875-0.99 to fp32 (lowest on the BN parameters of stage 1); HIP-vs-emulation 0.96-0.999.  So 0.98-to-fp32 is
    # not a property of this precision policy; what is asserted is that the kernels add nothing to the policy's own noise.
    # Round 4: the bar on "as aligned with fp32 as the emulation" is 0.04, not 0.02 - measured, not assumed: the two conv kernel
    # families (csrc/amp.hip ring kernel, csrc/amp_ws.hip) produce BIT-IDENTICAL convolution outputs (same k order) and differ
    # only in the order in which the BatchNorm partial sums are added, yet that alone moves the cosine-to-fp32 of the stage-1
    # BN parameters by up to 0.03 on this batch (layer1.1.bn1.bias: 0.929 / 0.899 / emulation 0.929; SSECG_AMP_WS = 0 / 1).  The
    # HIP-vs-emulation bar below (>= 0.93, measured 0.96) is the one a wrong kernel cannot pass.
    for k, ch, ce, chh in rows:
        assert ch >= ce - 0.04, f"{k}: cosine to fp32 {ch:.4f} < emulation's {ce:.4f} - 0.04"
        assert chh >= max(min(ch, ce) - 0.02, 0.93), f"{k}: HIP and emulation disagree ({chh:.4f}) more than either does with fp32"
    for k, ch, ce, chh in convs:
        assert ch >= 0.88, f"{k}: cosine to fp32 {ch:.4f} < 0.88"


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
    dxp = SAMP.to_planar(xb.grad).cpu()
    print(f"block {cin}->{cout} s{stride} L{L}: output rel. L2 {l2(o, out_ref):.2e}, input gradient {l2(dxp, xr.grad):.2e}, worst parameter "
          f"gradient {max(l2(prm.grad, sd['b.' + k].grad) for k, prm in blk.named_parameters()):.2e}")
    assert ((o - out_ref.detach()).abs().max() / out_ref.detach().abs().max()).item() < 2e-2
    assert l2(o, out_ref) < 2e-3, l2(o, out_ref)
    assert l2(dxp, xr.grad) < 5e-3, l2(dxp, xr.grad)
    for k, prm in blk.named_parameters():
        e = l2(prm.grad, sd["b." + k].grad)
        assert e < 1e-2, (k, e)
    for k, v in blk.state_dict().items():
        if "running" in k:
            assert rel(v, sd["b." + k]) < 1e-3, k


def test_amp_training_learns_like_fp32(dev):
    """60 FixMatch + AdamW steps on a LEARNABLE synthetic task (labels are a function of the signal) from the same init
    with the fp32 and the bf16 path: both must actually learn (supervised loss falls below half of ln 4, held-out accuracy
    far above the 25 % chance level, pseudo-labels start passing the 0.8 threshold) and the bf16 curve must track the fp32
    one.  A broken bf16 backward or optimiser path fails the "learns" half; label-independent data could not tell."""
    import algorithms.fixmatch as A_fm
    from utils.optimizer import get_optimizer_from_config
    C, B, L, seed, steps = 2, 16, 2000, 77, 60
    sd_np = synth.model_state(seed, C, trained=False)
    held = _learnable_batch(seed + 999, B, C, L)
    curves, acc, ratio = {}, {}, {}
    for amp in (False, True):
        model = build_hip_model(C, sd_np, dev)
        if amp:
            SAMP.enable(model)
        model.decode_head.dropout = None; model.decode_head.dropout_ratio = 0.0
        opt = get_optimizer_from_config(dict(TRAIN_CFG, lr=1e-3), model.parameters())
        hist = []
        for s in range(steps):
            b = to_dev({k: v for k, v in _learnable_batch(seed + 1 + s, B, C, L).items() if k != "u_target"}, dev)
            loss, stats = A_fm.fixmatch_step(model, b["labeled"]["ecg"], b["labeled"]["target"], b["unlabeled"]["ecg"],
                                             b["unlabeled"]["ecg_aug"], TRAIN_CFG["conf_thresh"])
            loss.backward()
            opt.step(); opt.zero_grad()
            hist.append(stats.clone())
        curves[amp] = torch.stack(hist).cpu().numpy()          # columns: loss_total, loss_x, loss_u_s, mask_ratio
        model.eval()
        with torch.no_grad():
            pred = SF.pseudo_label(model(torch.from_numpy(held["labeled"]["ecg"]).to(dev), return_loss=False)["seg_logits"])[1]
        acc[amp] = float((pred.cpu().numpy() == held["labeled"]["target"]).mean())
        ratio[amp] = float(curves[amp][-10:, 3].mean())
    a, b = curves[True], curves[False]
    assert np.isfinite(a).all() and np.isfinite(b).all()
    tail = slice(steps - 10, steps)
    print(f"loss_x first / last-10 mean: fp32 {b[0, 1]:.3f} {b[tail, 1].mean():.3f}  bf16 {a[0, 1]:.3f} {a[tail, 1].mean():.3f}; "
          f"held-out accuracy fp32 {acc[False]:.3f} bf16 {acc[True]:.3f}; mask_ratio (last 10) fp32 {ratio[False]:.2f} bf16 {ratio[True]:.2f}")
    assert abs(a[0, 1] - b[0, 1]) < 2e-2 * b[0, 1]
    # measured on MI355X: loss_x 1.447 -> 0.082 (both paths), held-out accuracy 0.974 / 0.975, mask_ratio 0.92 / 0.92
    for c in (a, b):
        assert c[tail, 1].mean() < 0.2 * np.log(4.0), "the supervised loss did not fall: nothing was learnt"
    assert acc[False] > 0.9 and acc[True] > 0.9
    assert abs(a[tail, 1].mean() - b[tail, 1].mean()) < 0.25 * b[tail, 1].mean() + 0.02
    assert abs(acc[True] - acc[False]) < 0.03
    assert ratio[False] > 0.5 and ratio[True] > 0.5


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


@pytest.mark.gpu
@pytest.mark.parametrize("N,C,Lin", [(6, 64, 1000), (3, 64, 52), (2, 16, 8), (5, 8, 4)])
def test_stem_pool_in_blocked_bf16_equals_the_separate_passes_bit_for_bit(N, C, Lin):
    """The stem's BN + ReLU + MaxPool(3, 2, 1) writing the pooled activation in blocked bf16 (round 4: no fp32 pooled tensor)
    against what it replaces - bn_relu_maxpool_fwd + to_blocked: the same comparison chains, the same rounding, so every output
    is identical bit for bit, NaN propagation included."""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(N * 1000 + Lin)
    c = torch.randn(N, C, Lin, generator=g).to(dev)
    c[0, 0, :4] = float("nan")                      # a NaN in a window propagates (nn.MaxPool1d) in both forms
    mean = (0.1 * torch.randn(C, generator=g)).to(dev); invstd = (0.5 + torch.rand(C, generator=g)).to(dev)
    gamma = (1.0 + 0.2 * torch.randn(C, generator=g)).to(dev); beta = (0.2 * torch.randn(C, generator=g)).to(dev)
    assert ops.stem_pool_b16_supported(N, C, Lin)
    yb = ops.stem_pool_fwd_b16(c, mean, invstd, gamma, beta)
    ref = SAMP.to_blocked(ops.bn_relu_maxpool_fwd(c, mean, invstd, gamma, beta, 3, 2, 1))
    assert yb.shape == ref.shape and torch.equal(yb.view(torch.int16), ref.view(torch.int16))


@pytest.mark.gpu
def test_amp_step_is_bit_identical_with_and_without_the_blocked_stem_pool(monkeypatch):
    """SSECG_AMP_STEM_BLOCKED=0 (ops.AMP_STEM_BLOCKED = False) keeps the fp32 pooled tensor and the two layout passes: the student
    pass - logits and all 65 gradients - must not change by a bit."""
    dev = torch.device("cuda:0")
    C, B, L, seed = 12, 8, 1000, 31
    sd_np = synth.model_state(seed, C, trained=True)
    batch = _learnable_batch(seed + 1, B, C, L)["labeled"]
    dm = torch.from_numpy(dropout_mask_np(seed + 1, B, lp=32)).to(dev, torch.uint8)     # L = 1000 -> 32 positions at the head
    x = torch.from_numpy(batch["ecg"]).to(dev); t = torch.from_numpy(batch["target"]).to(dev)
    outs = []
    for blocked in (True, False):
        monkeypatch.setattr(ops, "AMP_STEM_BLOCKED", blocked)
        model = _amp_model(C, sd_np, dev).train()
        model.decode_head.fixed_dropout_mask = dm
        logits = model(x, return_loss=False)["seg_logits"]
        F.cross_entropy(logits, t).backward()
        outs.append((logits.detach().clone(), {k: p.grad.detach().clone() for k, p in model.named_parameters()}))
    assert torch.equal(outs[0][0], outs[1][0])
    for k in outs[0][1]:
        assert torch.equal(outs[0][1][k], outs[1][1][k]), k


def test_amp_step_is_bit_identical_with_the_stem_tensors_stored_as_bf16(monkeypatch):
    """SSECG_AMP_STEM_C16 (ops.AMP_STEM_C16, default on): under use_amp the stem's conv output and its gradient - bf16-valued since the
    stem runs on 16-bit operands - are STORED as bf16 (L = 2000: 1000 conv positions, a multiple of 8); off = fp32 containers.  Logits and
    all 65 gradients must not change by a bit, and the switch must really change what is stored."""
    dev = torch.device("cuda:0")
    C, B, L, seed = 12, 4, 2000, 33
    sd_np = synth.model_state(seed, C, trained=True)
    batch = _learnable_batch(seed + 1, B, C, L)["labeled"]
    dm = torch.from_numpy(dropout_mask_np(seed + 1, B, lp=63)).to(dev, torch.uint8)     # L = 2000 -> 63 positions at the head
    x = torch.from_numpy(batch["ecg"]).to(dev); t = torch.from_numpy(batch["target"]).to(dev)
    outs, stored = [], []
    real = ops.stem_fwd_pair

    def spy(*a, **k):
        c, st = real(*a, **k)
        stored.append(c.dtype)
        return c, st

    monkeypatch.setattr(ops, "stem_fwd_pair", spy)
    for c16 in (True, False):
        monkeypatch.setattr(ops, "AMP_STEM_C16", c16)
        model = _amp_model(C, sd_np, dev).train()
        model.decode_head.fixed_dropout_mask = dm
        logits = model(x, return_loss=False)["seg_logits"]
        F.cross_entropy(logits, t).backward()
        outs.append((logits.detach().clone(), {k: p.grad.detach().clone() for k, p in model.named_parameters()}))
    assert stored == [torch.bfloat16, torch.float32]
    assert torch.equal(outs[0][0], outs[1][0])
    for k in outs[0][1]:
        assert torch.equal(outs[0][1][k], outs[1][1][k]), k
