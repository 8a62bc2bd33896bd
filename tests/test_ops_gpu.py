"""HIP kernels (through the C ABI) vs the torch-CPU oracle, op by op, fp32.

Tolerances: conv / BN results must agree to 2e-5 of the tensor's scale (the MFMA
path is an exact-fp32 FMA chain, only the summation order differs from oneDNN);
integer outputs (argmax masks, dropout masks, max-pool routing) bit-exact.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from ssecg import functional as SF
from ssecg import ops, synth

pytestmark = pytest.mark.gpu


def rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def rnd(seed, *shape, std=1.0):
    return torch.from_numpy(synth.normal(seed, 9, shape, std=std))


CONV_CASES = [
    # N, Cin, Lin, Cout, K, stride, pad, dil
    (2, 1, 2000, 64, 7, 2, 3, 1), (2, 2, 2000, 64, 7, 2, 3, 1), (2, 12, 2000, 64, 7, 2, 3, 1),
    (3, 64, 500, 64, 3, 1, 1, 1), (3, 64, 500, 128, 3, 2, 1, 1), (3, 64, 500, 128, 1, 2, 0, 1),
    (3, 128, 250, 256, 3, 2, 1, 1), (3, 128, 250, 256, 1, 2, 0, 1), (3, 256, 125, 256, 3, 1, 1, 1),
    (3, 256, 125, 512, 3, 2, 1, 1), (3, 256, 125, 512, 1, 2, 0, 1), (5, 512, 63, 512, 3, 1, 1, 1),
    (5, 512, 63, 128, 3, 1, 1, 1), (5, 128, 63, 4, 1, 1, 0, 1),
    (3, 5, 37, 7, 3, 1, 1, 1), (2, 3, 41, 33, 3, 2, 1, 1), (1, 70, 19, 130, 3, 1, 2, 2), (4, 9, 130, 65, 7, 2, 3, 1),
    # K smaller than one 16-deep stage (classifier dgrad Ktot=4, 1-lead stem Ktot=7): clamped weight reads
    (9, 128, 63, 4, 1, 1, 0, 1), (3, 1, 200, 64, 7, 2, 3, 1), (2, 4, 50, 130, 1, 1, 0, 1),
    # stride-2 data gradient as two parity phases: even / odd input lengths, tiles spanning sample boundaries
    (7, 64, 38, 64, 3, 2, 1, 1), (7, 64, 37, 48, 3, 2, 1, 1), (5, 128, 21, 256, 1, 2, 0, 1), (3, 48, 9, 80, 3, 2, 1, 1),
]


WINO_CASES = [
    # N, C, L, M: odd / even / tiny lengths, pair tiles spanning sample boundaries, both tile configs (M % 128 == 0 or not),
    # a single 8-channel stage, more pair tiles than workgroup slots is covered by the layer shapes in CONV_CASES
    (3, 64, 500, 64), (3, 256, 125, 256), (5, 512, 63, 512), (5, 512, 63, 128), (2, 8, 1, 64), (3, 8, 2, 64), (4, 16, 3, 128),
    (7, 24, 37, 192), (1, 64, 4096, 64), (9, 128, 31, 128), (130, 8, 5, 64),
    # the 64-channel F(4,3) tile (64 x 128 quads, round 4): odd lengths (masked vector tails), tiles spanning samples, three channel tiles
    (7, 64, 37, 64), (130, 64, 5, 64), (2, 128, 33, 64), (3, 64, 250, 192), (5, 64, 63, 64),
    # weight-gradient tiles (both channel counts multiples of 128): tiny / odd lengths, more slabs than pairs, 2 x 3 tiles
    (4, 128, 3, 128), (2, 128, 1, 128), (130, 128, 5, 128), (3, 384, 250, 256),
]


@pytest.fixture(params=[True, False], ids=["winograd", "direct"])
def wino(request, monkeypatch):
    monkeypatch.setattr(ops, "WINOGRAD", request.param)
    return request.param


@pytest.mark.parametrize("wino_f", [4, 2], ids=["default_F43_where_it_applies", "SSECG_WINO_F=2_F23_everywhere"])
@pytest.mark.parametrize("case", WINO_CASES)
def test_conv_winograd_f23(case, wino_f, dev, monkeypatch):
    """3-tap stride-1 convs in Winograd form (forward with BN statistics / folded epilogue, data gradient with accumulation)
    against F.conv1d: F(4,3) where both channel counts are multiples of 64 and F(2,3) elsewhere (default), and F(2,3) on every
    shape (ops.WINO_F = 2, the SSECG_WINO_F=2 switch)."""
    monkeypatch.setattr(ops, "WINO_F", wino_f)
    N, C, L, M = case
    assert ops.WINOGRAD and lib_supported(N, C, L, M)
    x = rnd(1, N, C, L).requires_grad_(True)
    w = rnd(2, M, C, 3, std=math.sqrt(2.0 / (3 * M))).requires_grad_(True)
    y_ref = F.conv1d(x, w, padding=1)
    dy = rnd(3, *y_ref.shape)
    dx_ref, dw_ref = torch.autograd.grad(y_ref, (x, w), dy)
    xg, wg, dyg = x.detach().to(dev), w.detach().to(dev), dy.to(dev)
    before = ops.WINO_TRANSFORMS[0]
    y, stats = ops.conv1d_fwd(xg, wg, 1, 1, 1, want_stats=True)
    assert ops.WINO_TRANSFORMS[0] == before + 1                   # took the Winograd path
    assert rel(y, y_ref) < 2e-5
    sums = ops.bn_reduce_partials(stats).cpu()
    ref_s = y_ref.detach().double().sum(dim=(0, 2)); ref_q = (y_ref.detach().double() ** 2).sum(dim=(0, 2))
    assert ((sums[:, 0] - ref_s).abs().max() / (ref_q.sqrt().max() + 1e-30)).item() < 1e-4
    assert rel(sums[:, 1], ref_q) < 2e-5
    sc = 1.0 + 0.2 * rnd(5, M); sh = 0.3 * rnd(6, M); res = rnd(7, *y_ref.shape)
    ye, _ = ops.conv1d_fwd(xg, wg, 1, 1, 1, scale=sc.to(dev), shift=sh.to(dev), residual=res.to(dev), relu=True)
    assert rel(ye, F.relu(y_ref.detach() * sc[None, :, None] + sh[None, :, None] + res)) < 2e-5
    dx = ops.conv1d_dgrad(dyg, wg, L, 1, 1, 1)
    assert rel(dx, dx_ref) < 2e-5
    acc = rnd(4, N, C, L)
    dx2 = ops.conv1d_dgrad(dyg, wg, L, 1, 1, 1, accumulate=acc.to(dev))
    assert rel(dx2, dx_ref + acc) < 2e-5
    from ssecg.lib import lib
    assert (lib().ssecg_conv1d_wino_wgrad_supported(N, C, L, M) == 1) == (C % 128 == 0 and M % 128 == 0)
    assert (lib().ssecg_conv1d_wino_wgrad4_supported(N, C, L, M) == 1) == (C % 64 == 0 and M % 64 == 0)
    for wf in (4, 2):                                             # transpose of F(4,3) (default) / of F(2,3) (SSECG_WINO_WGRAD_F=2)
        monkeypatch.setattr(ops, "WINO_WGRAD_F", wf)
        wino_wgrad = C % (64 if wf == 4 else 128) == 0 and M % (64 if wf == 4 else 128) == 0
        dw = ops.conv1d_wgrad(dyg, xg, 3, 1, 1, 1)                # Winograd form when supported, direct otherwise
        assert rel(dw, dw_ref) < 2e-5, wf
        if wino_wgrad:
            assert torch.equal(dw, ops.conv1d_wgrad(dyg, xg, 3, 1, 1, 1))   # fixed slab order: bitwise reproducible
    # standalone op calls never trust a cached operand: weights rewritten through ``.data`` (its own version counter - the
    # reference rebinds .data, src/algorithms/mean_teacher.py:144) are seen without any manual call
    wg.data.mul_(2.0)
    assert rel(ops.conv1d_fwd(xg, wg, 1, 1, 1)[0], 2.0 * y_ref) < 2e-5
    assert rel(ops.conv1d_dgrad(dyg, wg, L, 1, 1, 1), 2.0 * dx_ref) < 2e-5


def test_wino_f64_switch_keeps_the_64_channel_layer_on_f23(dev, monkeypatch):
    """SSECG_WINO_F64=2: the 64-channel convolutions go back to the 16-wave F(2,3) kernel (the A/B switch of the round-4 change);
    both selections agree with F.conv1d and with each other at the kernel bar."""
    x = rnd(1, 6, 64, 500)
    w = rnd(2, 64, 64, 3, std=math.sqrt(2.0 / (3 * 64)))
    ref = F.conv1d(x, w, padding=1)
    xg, wg = x.to(dev), w.to(dev)
    assert ops._wino_variant(64, 64) == 4 and ops._wino_symbol(64, var=4) == "conv_wino4_kernel<2, 4>"
    y4, _ = ops.conv1d_fwd(xg, wg, 1, 1, 1)
    monkeypatch.setattr(ops, "WINO_F64", 2)
    assert ops._wino_variant(64, 64) == 2 and ops._wino_variant(128, 128) == 4
    y2, _ = ops.conv1d_fwd(xg, wg, 1, 1, 1)
    assert rel(y4, ref) < 2e-5 and rel(y2, ref) < 2e-5 and rel(y4, y2) < 2e-5 and not torch.equal(y4, y2)


@pytest.mark.parametrize("case", [(16, 512, 63, 512), (32, 256, 125, 256), (16, 128, 250, 128), (3, 512, 63, 128), (64, 512, 63, 512),
                                  (128, 512, 63, 512)])
def test_wino4_k_split_of_small_launches(case, dev, monkeypatch):
    """Small F(4,3) launches (fewer tiles than CUs) contract their channels in K splits side by side and finish with one summing
    pass that applies the epilogue and takes the BatchNorm sums: against torch and against the unsplit kernel - eval forward
    (folded BN + residual + ReLU), train forward (raw output + statistics, with and without the producer's BN + ReLU fused into
    the gather) and data gradient (with the accumulated residual-branch gradient)."""
    from ssecg.lib import lib
    monkeypatch.setattr(ops, "KSPLIT", True)             # the default (SSECG_KSPLIT=0 switches it off)
    N, C, L, M = case
    S = lib().ssecg_conv1d_wino4_split(N, C, L, M)
    qtiles = -(-(N * -(-L // 4)) // 64)
    tiles = qtiles * (M // 128)
    assert (S > 1) == (tiles <= 128 and qtiles <= 64) and (C // S) % 16 == 0 and tiles * S <= 256   # never more workgroup columns than the 256 CUs hold
    assert lib().ssecg_conv1d_wino4_split(512, 512, 63, 128) == 1    # the teacher pass's head conv at the bench size: 128 tiles, not split
    assert lib().ssecg_conv1d_wino4_split(1024, C, L, M) == 1        # the bench's batch never splits
    x = rnd(1, N, C, L)
    w = rnd(2, M, C, 3, std=math.sqrt(2.0 / (3 * M)))
    sc, sh, res = 1.0 + 0.2 * rnd(3, M), 0.3 * rnd(4, M), rnd(5, N, M, L)
    ref = F.relu(F.conv1d(x, w, padding=1) * sc[None, :, None] + sh[None, :, None] + res)
    xg, wg, scg, shg, resg = x.to(dev), w.to(dev), sc.to(dev), sh.to(dev), res.to(dev)
    isc, ish = (1.0 + 0.2 * rnd(8, C)).to(dev), (0.3 * rnd(9, C)).to(dev)

    def run():
        y, _ = ops.conv1d_fwd(xg, wg, 1, 1, 1, scale=scg, shift=shg, residual=resg, relu=True)
        yt, st = ops.conv1d_fwd(xg, wg, 1, 1, 1, want_stats=True)
        ya, sa = ops.conv1d_fwd(xg, wg, 1, 1, 1, want_stats=True, in_affine=(isc, ish))
        dx = ops.conv1d_dgrad(dy.to(dev), wg, L, 1, 1, 1, accumulate=acc.to(dev))
        return y, yt, ops.bn_reduce_partials(st), ya, ops.bn_reduce_partials(sa), dx

    dy, acc = rnd(6, N, M, L), rnd(7, N, C, L)
    y, yt, st, ya, sa, dx = run()
    assert rel(y, ref) < 2e-5
    yt_ref = F.conv1d(x, w, padding=1)
    assert rel(yt, yt_ref) < 2e-5
    sums_ref = torch.stack([yt_ref.double().sum(dim=(0, 2)), (yt_ref.double() ** 2).sum(dim=(0, 2))], dim=1)
    assert rel(st.view(-1, 2)[:, 1], sums_ref[:, 1]) < 2e-5
    assert ((st.view(-1, 2)[:, 0].cpu() - sums_ref[:, 0]).abs().max() / sums_ref[:, 1].sqrt().max()).item() < 1e-4
    ya_ref = F.conv1d(F.relu(x * isc.cpu()[None, :, None] + ish.cpu()[None, :, None]), w, padding=1)
    assert rel(ya, ya_ref) < 2e-5
    assert rel(sa.view(-1, 2)[:, 1], (ya_ref.double() ** 2).sum(dim=(0, 2))) < 2e-5
    dx_ref = torch.nn.grad.conv1d_input((N, C, L), w, dy, padding=1) + acc
    assert rel(dx, dx_ref) < 2e-5
    monkeypatch.setattr(ops, "KSPLIT", False)
    y0, yt0, st0, ya0, sa0, dx0 = run()
    assert rel(y, y0) < 1e-5 and rel(dx, dx0) < 1e-5 and rel(yt, yt0) < 1e-5 and rel(ya, ya0) < 1e-5
    assert rel(st.view(-1, 2)[:, 1], st0.view(-1, 2)[:, 1]) < 2e-5 and rel(sa.view(-1, 2)[:, 1], sa0.view(-1, 2)[:, 1]) < 2e-5
    if S == 1:
        assert torch.equal(y, y0) and torch.equal(dx, dx0) and torch.equal(yt, yt0) and torch.equal(st, st0) and torch.equal(ya, ya0)
    else:
        assert not torch.equal(yt, yt0)                   # the split really ran (another summation order)


@pytest.mark.parametrize("case", [(32, 256, 125, 512, 3, 2, 1), (32, 128, 250, 256, 3, 2, 1), (16, 256, 125, 512, 1, 2, 0), (32, 64, 500, 128, 3, 2, 1),
                                  (8, 512, 63, 128, 1, 1, 0), (64, 256, 125, 512, 3, 2, 1), (5, 256, 125, 512, 3, 2, 1)])
def test_igemm_k_split_of_small_launches(case, dev, monkeypatch):
    """The implicit-GEMM launches of small batches (stride-2 three-tap and 1x1 convolutions, their phase data gradients) split
    their contraction the same way (round 5): against torch and against the unsplit kernel - eval forward, train forward with
    statistics and the fused input BN, data gradient with an accumulated gradient, the in-place 1x1 stride-2 data gradient."""
    from ssecg.lib import lib
    N, Cin, Lin, Cout, K, stride, pad = case
    Lout = (Lin + 2 * pad - K) // stride + 1
    x = rnd(1, N, Cin, Lin)
    w = rnd(2, Cout, Cin, K, std=math.sqrt(2.0 / (K * Cout)))
    sc, sh, res = 1.0 + 0.2 * rnd(3, Cout), 0.3 * rnd(4, Cout), rnd(5, N, Cout, Lout)
    dy, acc = rnd(6, N, Cout, Lout), rnd(7, N, Cin, Lin)
    xg, wg, scg, shg, resg, dyg, accg = (t.to(dev) for t in (x, w, sc, sh, res, dy, acc))
    isc, ish = (1.0 + 0.2 * rnd(8, Cin)).to(dev), (0.3 * rnd(9, Cin)).to(dev)
    nb_f = lib().ssecg_conv1d_fwd_split_workspace(N, Cin, Lin, Cout, Lout, K)
    nb_d = lib().ssecg_conv1d_dgrad_split_workspace(N, Cin, Lin, Cout, Lout, K, stride)
    assert lib().ssecg_conv1d_fwd_split_workspace(1024, Cin, Lin, Cout, Lout, K) == 0      # the bench's batch never splits

    def run():
        y, _ = ops.conv1d_fwd(xg, wg, stride, pad, 1, scale=scg, shift=shg, residual=resg, relu=True)
        yt, st = ops.conv1d_fwd(xg, wg, stride, pad, 1, want_stats=True)
        ya, sa = ops.conv1d_fwd(xg, wg, stride, pad, 1, want_stats=True, in_affine=(isc, ish))
        dx = ops.conv1d_dgrad(dyg, wg, Lin, stride, pad, 1, accumulate=(accg if not (K == 1 and stride == 2) else None))
        dxi = None
        if K == 1 and stride == 2:                       # the downsample branch's gradient added in place at the even positions
            dxi = ops.conv1d_dgrad(dyg, wg, Lin, stride, pad, 1, accumulate=accg.clone(), inplace=True)
        return y, yt, ops.bn_reduce_partials(st), ya, ops.bn_reduce_partials(sa), dx, dxi

    monkeypatch.setattr(ops, "KSPLIT", True)
    y, yt, st, ya, sa, dx, dxi = run()
    yt_ref = F.conv1d(x, w, stride=stride, padding=pad)
    assert rel(y, F.relu(yt_ref * sc[None, :, None] + sh[None, :, None] + res)) < 2e-5 and rel(yt, yt_ref) < 2e-5
    assert rel(st.view(-1, 2)[:, 1], (yt_ref.double() ** 2).sum(dim=(0, 2))) < 2e-5
    ya_ref = F.conv1d(F.relu(x * isc.cpu()[None, :, None] + ish.cpu()[None, :, None]), w, stride=stride, padding=pad)
    assert rel(ya, ya_ref) < 2e-5 and rel(sa.view(-1, 2)[:, 1], (ya_ref.double() ** 2).sum(dim=(0, 2))) < 2e-5
    dx_ref = torch.nn.grad.conv1d_input((N, Cin, Lin), w, dy, stride=stride, padding=pad)
    if K == 1 and stride == 2:
        assert rel(dx, dx_ref) < 2e-5 and rel(dxi, dx_ref + acc) < 2e-5
    else:
        assert rel(dx, dx_ref + acc) < 2e-5
    monkeypatch.setattr(ops, "KSPLIT", False)
    y0, yt0, st0, ya0, sa0, dx0, dxi0 = run()
    for a, b_ in ((y, y0), (yt, yt0), (ya, ya0), (dx, dx0)) + (((dxi, dxi0),) if dxi is not None else ()):
        assert rel(a, b_) < 1e-5
    if nb_f == 0:
        assert torch.equal(yt, yt0) and torch.equal(y, y0) and torch.equal(st, st0)
    else:
        assert not torch.equal(yt, yt0)
    if nb_d == 0:
        assert torch.equal(dx, dx0)


def lib_supported(N, C, L, M):
    from ssecg.lib import lib
    return (lib().ssecg_conv1d_wino4_supported if ops._wino_variant(M, C) == 4 else lib().ssecg_conv1d_wino_supported)(N, C, L, M) == 1


def _random_conv_shapes(n=48, seed=20261004):
    """Seeded sweep over the shape space the hand-picked CONV_CASES sample: every kernel family (stem-like, direct, F(2,3), F(4,3),
    generic fallback), odd / tiny lengths, batch sizes that leave partial tiles, strides 1-2, taps 1 / 3 / 7, dilation."""
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        fam = int(rng.integers(0, 6))
        N = int(rng.integers(1, 9)) if fam != 5 else int(rng.integers(20, 70))
        if fam == 0:      # stem-like
            c = (N, int(rng.integers(1, 13)), int(rng.integers(20, 400)), 64, 7, 2, 3, 1)
        elif fam == 1:    # F(4,3): both channel counts multiples of 128
            c = (N, 128 * int(rng.integers(1, 5)), int(rng.integers(1, 140)), 128 * int(rng.integers(1, 5)), 3, 1, 1, 1)
        elif fam == 2:    # F(2,3) / direct fast path: multiples of 16 (8 for F(2,3))
            c = (N, 8 * int(rng.integers(1, 20)), int(rng.integers(1, 300)), 16 * int(rng.integers(1, 12)), 3, 1, 1, 1)
        elif fam == 3:    # stride 2, 3 taps and 1 tap
            k = int(rng.choice([1, 3]))
            c = (N, 16 * int(rng.integers(1, 17)), int(rng.integers(2, 200)), 16 * int(rng.integers(1, 17)), k, 2, k // 2, 1)
        elif fam == 4:    # generic: odd channel counts, dilation
            d = int(rng.integers(1, 3))
            c = (N, int(rng.integers(1, 40)), int(rng.integers(5, 80)), int(rng.integers(1, 70)), 3, int(rng.integers(1, 3)), d, d)
        else:             # many samples, short rows: tiles span many samples
            c = (N, 64, int(rng.integers(1, 12)), 64, 3, 1, 1, 1)
        if conv_len(c[2], c[4], c[5], c[6], c[7]) >= 1:
            out.append(c)
    return out


def conv_len(lin, k, s, p, d):
    return (lin + 2 * p - d * (k - 1) - 1) // s + 1


@pytest.mark.parametrize("case", _random_conv_shapes(), ids=lambda c: "x".join(map(str, c)))
def test_conv_random_shapes(case, dev):
    """Forward (+ BN partial sums), data gradient and weight gradient of a seeded random sweep of shapes against torch at 2e-5."""
    N, Cin, Lin, Cout, K, s, p, d = case
    x = rnd(11, N, Cin, Lin).requires_grad_(True)
    w = rnd(12, Cout, Cin, K, std=math.sqrt(2.0 / (K * Cout))).requires_grad_(True)
    y_ref = F.conv1d(x, w, stride=s, padding=p, dilation=d)
    dy = rnd(13, *y_ref.shape)
    dx_ref, dw_ref = torch.autograd.grad(y_ref, (x, w), dy)
    xg, wg, dyg = x.detach().to(dev), w.detach().to(dev), dy.to(dev)
    y, stats = ops.conv1d_fwd(xg, wg, s, p, d, want_stats=True)
    assert rel(y, y_ref) < 2e-5
    sums = ops.bn_reduce_partials(stats).cpu()
    yd = y_ref.detach().double()
    scale = max(1.0, float((yd ** 2).sum(dim=(0, 2)).max()))
    assert (sums[:, 0] - yd.sum(dim=(0, 2))).abs().max() < 2e-5 * scale and (sums[:, 1] - (yd ** 2).sum(dim=(0, 2))).abs().max() < 2e-5 * scale
    assert rel(ops.conv1d_dgrad(dyg, wg, Lin, s, p, d), dx_ref) < 2e-5
    assert rel(ops.conv1d_wgrad(dyg, xg, K, s, p, d), dw_ref) < 2e-5


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(case, dev, wino):
    N, Cin, Lin, Cout, K, s, p, d = case
    if not wino and not (K == 3 and s == 1 and p == 1 and d == 1 and lib_supported(N, Cin, Lin, Cout)):
        pytest.skip("not a Winograd shape: identical to the winograd=True run")
    x = rnd(1, N, Cin, Lin).requires_grad_(True)
    w = rnd(2, Cout, Cin, K, std=math.sqrt(2.0 / (K * Cout))).requires_grad_(True)
    y_ref = F.conv1d(x, w, stride=s, padding=p, dilation=d)
    dy = rnd(3, *y_ref.shape)
    dx_ref, dw_ref = torch.autograd.grad(y_ref, (x, w), dy)
    xg, wg, dyg = x.detach().to(dev), w.detach().to(dev), dy.to(dev)
    y, stats = ops.conv1d_fwd(xg, wg, s, p, d, want_stats=True)
    assert rel(y, y_ref) < 2e-5
    sums = ops.bn_reduce_partials(stats).cpu()
    ref_s = y_ref.detach().double().sum(dim=(0, 2)); ref_q = (y_ref.detach().double() ** 2).sum(dim=(0, 2))
    assert ((sums[:, 0] - ref_s).abs().max() / (ref_q.sqrt().max() + 1e-30)).item() < 1e-4
    assert rel(sums[:, 1], ref_q) < 2e-5
    dx = ops.conv1d_dgrad(dyg, wg, Lin, s, p, d)
    assert rel(dx, dx_ref) < 2e-5
    acc = rnd(4, N, Cin, Lin)
    dx2 = ops.conv1d_dgrad(dyg, wg, Lin, s, p, d, accumulate=acc.to(dev))
    assert rel(dx2, dx_ref + acc) < 2e-5
    dw = ops.conv1d_wgrad(dyg, xg, K, s, p, d)
    assert rel(dw, dw_ref) < 2e-5


@pytest.mark.parametrize("case", [(3, 64, 500, 64, 3, 1, 1, 1), (5, 256, 125, 512, 3, 2, 1, 1), (3, 5, 37, 7, 3, 1, 1, 1),
                                  (5, 128, 63, 4, 1, 1, 0, 1)])
def test_conv_fwd_epilogue(case, dev):
    N, Cin, Lin, Cout, K, s, p, d = case
    x = rnd(1, N, Cin, Lin); w = rnd(2, Cout, Cin, K, std=0.1)
    sc = 1.0 + 0.2 * rnd(5, Cout); sh = 0.3 * rnd(6, Cout)
    y0 = F.conv1d(x, w, stride=s, padding=p, dilation=d)
    res = rnd(7, *y0.shape)
    ref = F.relu(y0 * sc[None, :, None] + sh[None, :, None] + res)
    y, _ = ops.conv1d_fwd(x.to(dev), w.to(dev), s, p, d, scale=sc.to(dev), shift=sh.to(dev), residual=res.to(dev), relu=True)
    assert rel(y, ref) < 2e-5
    yb, _ = ops.conv1d_fwd(x.to(dev), w.to(dev), s, p, d, shift=sh.to(dev))
    assert rel(yb, y0 + sh[None, :, None]) < 2e-5


def _random_bn_shapes(n=16, seed=4242):
    rng = np.random.default_rng(seed)
    return [(int(rng.integers(1, 40)), int(rng.choice([1, 3, 8, 24, 64, 96, 128, 200])), int(rng.integers(1, 300))) for _ in range(n)]


@pytest.mark.parametrize("shape", [(4, 64, 500), (3, 256, 125), (5, 512, 63), (2, 7, 37)] + _random_bn_shapes())
@pytest.mark.parametrize("relu,use_res", [(True, False), (True, True), (False, False)])
def test_bn_train_fwd_bwd(shape, relu, use_res, dev):
    N, C, L = shape
    x = (rnd(1, N, C, L) * 1.7 + 0.4).requires_grad_(True)
    g = (1.0 + 0.2 * rnd(2, C)).requires_grad_(True); b = (0.1 * rnd(3, C)).requires_grad_(True)
    res = rnd(4, N, C, L).requires_grad_(True) if use_res else None
    rm, rv = 0.2 * rnd(5, C), 1.0 + 0.5 * rnd(6, C).abs()
    rm_ref, rv_ref = rm.clone(), rv.clone()
    y_ref = F.batch_norm(x, rm_ref, rv_ref, g, b, training=True, momentum=0.1, eps=1e-5)
    if use_res: y_ref = y_ref + res
    if relu: y_ref = F.relu(y_ref)
    dy = rnd(7, N, C, L)
    grads = torch.autograd.grad(y_ref, (x, g, b) + ((res,) if use_res else ()), dy)
    # HIP: statistics from exact per-channel sums of x (here produced by a 1x1 identity-free path: direct partials)
    xg = x.detach().to(dev)
    partial = torch.stack([xg.double().sum(dim=(0, 2)).float(), (xg.double() ** 2).sum(dim=(0, 2)).float()], dim=1)[None].contiguous()
    sums = ops.bn_reduce_partials(partial)
    rmg, rvg = rm.to(dev), rv.to(dev)
    mean, invstd = ops.bn_finalize(sums, N * L, 1e-5, 0.1, rmg, rvg)
    assert rel(rmg, rm_ref) < 1e-5 and rel(rvg, rv_ref) < 1e-5
    gg, bg = g.detach().to(dev), b.detach().to(dev)
    resg = res.detach().to(dev) if use_res else None
    y = ops.bn_apply_fwd(xg, mean, invstd, gg, bg, resg, relu)
    assert rel(y, y_ref) < 1e-5
    dyg = dy.to(dev)
    part = ops.bn_bwd_reduce(dyg, y if relu else None, xg, mean, invstd)
    s2 = ops.bn_reduce_partials(part)
    dgam, dbet = ops.bn_param_grads(s2)
    dx, dz = ops.bn_bwd_apply(dyg, y if relu else None, xg, mean, invstd, gg, s2, N * L, want_dz=use_res)
    if relu and not use_res:  # ReLU mask recomputed from the BN input instead of read from the saved activation
        part_r = ops.bn_bwd_reduce(dyg, None, xg, mean, invstd, gg, bg, relu_recompute=True)
        assert rel(part_r, part) < 1e-6
        dx_r, _ = ops.bn_bwd_apply(dyg, None, xg, mean, invstd, gg, s2, N * L, beta=bg, relu_recompute=True)
        assert rel(dx_r, dx) < 1e-6 and rel(dx_r, grads[0]) < 3e-5
    assert rel(dx, grads[0]) < 3e-5
    # dgamma / dbeta are sums of N * L signed terms: where they cancel (a single channel whose sum of 1917 O(1) terms is 0.3) the
    # fp32 reference itself sits 9e-6 from the fp64 value, and any other summation order lands as far on the other side - the bar
    # is 3e-5 of the largest gradient plus a third of an fp32 ulp of the sum of the terms' magnitudes
    with torch.no_grad():
        dzr = dy * (y_ref > 0) if relu else dy
        xh = (x - x.mean(dim=(0, 2), keepdim=True)) * torch.rsqrt(x.var(dim=(0, 2), unbiased=False, keepdim=True) + 1e-5)
        mag_g, mag_b = (dzr * xh).abs().sum(dim=(0, 2)), dzr.abs().sum(dim=(0, 2))
    assert ((dgam.cpu() - grads[1]).abs() <= 3e-5 * grads[1].abs().max() + 2e-8 * mag_g).all()
    assert ((dbet.cpu() - grads[2]).abs() <= 3e-5 * grads[2].abs().max() + 2e-8 * mag_b).all()
    if use_res:
        assert rel(dz, grads[3]) < 1e-6
    if relu and ops.bn_mask_supported(N, C, L):
        # packed ReLU mask written by the apply pass: bit (e & 7) of byte (e >> 3) = (y[e] > 0); both backward passes driven
        # by it give bit-identical results to the ones driven by the saved activation
        y2, mask = ops.bn_apply_fwd(xg, mean, invstd, gg, bg, resg, relu, want_mask=True)
        assert torch.equal(y2, y)
        bits = np.unpackbits(mask.cpu().numpy(), bitorder="little")[: N * C * L]
        assert np.array_equal(bits.astype(bool), (y.cpu().numpy() > 0).reshape(-1))
        assert torch.equal(ops.bn_bwd_reduce(dyg, mask, xg, mean, invstd), part)
        dx_m, dz_m = ops.bn_bwd_apply(dyg, mask, xg, mean, invstd, gg, s2, N * L, want_dz=use_res)
        assert torch.equal(dx_m, dx) and (not use_res or torch.equal(dz_m, dz))
    elif relu:
        with pytest.raises(Exception):
            ops.bn_apply_fwd(xg, mean, invstd, gg, bg, resg, relu, want_mask=True)


def test_bn_fold(dev):
    C = 96
    g, b, rm, rv = 1 + 0.2 * rnd(1, C), 0.1 * rnd(2, C), 0.3 * rnd(3, C), 1 + rnd(4, C).abs()
    sc, sh = ops.bn_fold(g.to(dev), b.to(dev), rm.to(dev), rv.to(dev), 1e-5)
    x = rnd(5, 2, C, 33)
    ref = F.batch_norm(x, rm, rv, g, b, training=False, eps=1e-5)
    assert rel(x.to(dev) * sc[None, :, None] + sh[None, :, None], ref) < 1e-6


@pytest.mark.parametrize("L", [1000, 125, 63, 7])
def test_maxpool_fwd_bwd_with_ties(L, dev):
    x = F.relu(rnd(1, 3, 8, L)).requires_grad_(True)  # post-ReLU input: many exact ties at 0
    y_ref = F.max_pool1d(x, 3, 2, 1)
    dy = rnd(2, *y_ref.shape)
    (dx_ref,) = torch.autograd.grad(y_ref, x, dy)
    y = ops.maxpool1d_fwd(x.detach().to(dev))
    assert torch.equal(y.cpu(), y_ref.detach())
    dx = ops.maxpool1d_bwd(x.detach().to(dev), dy.to(dev))
    assert rel(dx, dx_ref) < 1e-6


@pytest.mark.parametrize("lin,lout,align", [(63, 2000, False), (63, 2000, True), (79, 2500, False), (7, 20, False),
                                             (50, 20, False), (1, 9, False)])
def test_interp_fwd_bwd(lin, lout, align, dev):
    x = rnd(1, 3, 4, lin).requires_grad_(True)
    y_ref = F.interpolate(x, size=lout, mode="linear", align_corners=align)
    dy = rnd(2, 3, 4, lout)
    (dx_ref,) = torch.autograd.grad(y_ref, x, dy)
    y = ops.interp_linear_fwd(x.detach().to(dev), lout, align)
    assert rel(y, y_ref) < 1e-5
    dx = ops.interp_linear_bwd(dy.to(dev), lin, align)
    assert rel(dx, dx_ref) < 1e-5


def test_dropout(dev):
    x = rnd(1, 6, 128, 63).to(dev)
    y, m = ops.dropout_fwd(x, 0.1, 1234)
    keep = m.float().mean().item()
    assert abs(keep - 0.9) < 0.01
    assert torch.equal(y, torch.where(m.bool(), x * (1.0 / 0.9), torch.zeros_like(x)))
    y2, m2 = ops.dropout_fwd(x, 0.1, 1234)
    assert torch.equal(m, m2)
    _, m3 = ops.dropout_fwd(x, 0.1, 1235)
    assert not torch.equal(m, m3)
    assert torch.equal(ops.mask_scale(x, m, 1.0 / 0.9), y)


@pytest.mark.parametrize("K", [4, 6])
def test_pseudo_label_and_losses(K, dev):
    N, L = 3, 2000
    logits = (rnd(1, N, K, L) * 3).requires_grad_(True)
    conf_ref = logits.softmax(1).max(1)[0]; mask_ref = logits.argmax(1)
    conf, mask, prob = ops.softmax_conf_argmax(logits.detach().to(dev), want_prob=True)
    assert torch.equal(mask.cpu(), mask_ref)
    assert rel(conf, conf_ref) < 1e-6 and rel(prob, logits.softmax(1)) < 1e-6
    tgt = torch.from_numpy(synth.labels(2, 4, N, L, K))
    # plain CE
    loss_ref = F.cross_entropy(logits, tgt)
    (g_ref,) = torch.autograd.grad(loss_ref, logits)
    dl, part = ops.ce_hard_fwd_bwd(logits.detach().to(dev), tgt.to(dev), None, 0.0, 1.0 / (N * L))
    out = ops.sum_partials(part, 1.0 / (N * L)).cpu()
    assert abs(out[0].item() - loss_ref.item()) < 1e-6 * max(1, abs(loss_ref.item())) and abs(out[1].item() - 1.0) < 1e-6
    assert rel(dl, g_ref) < 1e-5
    # masked CE (FixMatch)
    cf = torch.from_numpy(synth.uniform(3, 5, N * L).astype(np.float32)).reshape(N, L)
    keep = cf >= 0.8
    loss_ref = (F.cross_entropy(logits, tgt, reduction="none") * keep).mean()
    (g_ref,) = torch.autograd.grad(loss_ref, logits)
    dl, part = ops.ce_hard_fwd_bwd(logits.detach().to(dev), tgt.to(dev), cf.to(dev), 0.8, 1.0 / (N * L))
    out = ops.sum_partials(part, 1.0 / (N * L)).cpu()
    assert abs(out[0].item() - loss_ref.item()) < 1e-6 and abs(out[1].item() - keep.float().mean().item()) < 1e-6
    assert rel(dl, g_ref) < 1e-5
    # soft CE (MeanTeacher)
    p = (rnd(4, N, K, L) * 2).softmax(1)
    loss_ref = F.cross_entropy(logits, p)
    (g_ref,) = torch.autograd.grad(loss_ref, logits)
    dl, part = ops.ce_soft_fwd_bwd(logits.detach().to(dev), p.to(dev), 1.0 / (N * L))
    out = ops.sum_partials(part, 1.0 / (N * L)).cpu()
    assert abs(out[0].item() - loss_ref.item()) < 2e-6 * max(1, abs(loss_ref.item()))
    assert rel(dl, g_ref) < 1e-5


@pytest.mark.parametrize("shape", [(5, 4, 63), (1024, 4, 63), (1, 4, 7), (33, 128, 50), (3, 300, 20)])
def test_channel_sum(shape, dev):
    """Per-channel sums (the classifier's bias gradient): one sample, fewer samples than slabs, the bench's shape, and more
    channels than the slab scratch holds (single-workgroup path); slab partials are combined in a fixed order, so repeated
    launches agree bit for bit."""
    x = rnd(1, *shape)
    xg = x.to(dev)
    got = ops.channel_sum(xg)
    assert rel(got, x.double().sum(dim=(0, 2))) < 1e-5
    for _ in range(3):
        assert torch.equal(ops.channel_sum(xg), got)


def test_adamw_and_ema_multi_tensor(dev):
    """HIP multi-tensor AdamW / EMA vs the oracle's single-tensor restatement on identical gradients."""
    from oracle import torch_ref as O
    from ssecg.optim import EmaUpdater, FusedAdamW
    shapes = [(64, 12, 7), (64,), (128, 64, 3), (4,), (512, 512, 3), (1,)]
    ps = [torch.nn.Parameter(rnd(10 + i, *s).to(dev)) for i, s in enumerate(shapes)]
    sd = {f"p{i}": p.detach().cpu().clone() for i, p in enumerate(ps)}
    opt = FusedAdamW(ps, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05)
    ostate = {}
    for step in range(3):
        lr = 3e-4 * (step + 1)
        grads = {f"p{i}": rnd(100 + 10 * step + i, *s) * (10.0 ** (i - 3)) for i, s in enumerate(shapes)}
        for i, p in enumerate(ps):
            p.grad = grads[f"p{i}"].to(dev)
        for g_ in opt.param_groups:
            g_["lr"] = lr
        opt.step()
        O.adamw_step(sd, grads, ostate, lr, (0.9, 0.999), 1e-8, 0.05)
        for i, p in enumerate(ps):
            assert rel(p, sd[f"p{i}"]) < 2e-6, (step, i)
            assert rel(opt.state[p]["exp_avg_sq"], ostate[f"exp_avg_sq.p{i}"]) < 2e-6
    assert int(opt.state[ps[0]]["step"].item()) == 3
    # EMA over parameters and buffers, incl. an int64 counter turning float32 and initial aliasing (Q4)
    class M(torch.nn.Module):
        def __init__(self, seed):
            super().__init__()
            self.w = torch.nn.Parameter(rnd(seed, 300, 7))
            self.register_buffer("rm", rnd(seed + 1, 33))
            self.register_buffer("nbt", torch.tensor(5 + seed, dtype=torch.int64))
    st, te = M(1).to(dev), M(2).to(dev)
    te.w.data = st.w.data  # aliased at construction
    ref_t = {"w": st.w.detach().cpu().clone(), "rm": te.rm.cpu().clone(), "nbt": te.nbt.cpu().clone()}
    ema = EmaUpdater()
    for it in range(2):
        with torch.no_grad():
            st.w.add_(0.01 * (it + 1)); st.nbt.add_(1)
        ref_s = {"w": st.w.detach().cpu(), "rm": st.rm.cpu(), "nbt": st.nbt.cpu()}
        if it == 0:
            ref_t["w"] = ref_s["w"]  # the aliased teacher saw the student's in-place update (Q4)
        ema(st, te, 0.99)
        O.ema_update(ref_s, ref_t, 0.99)
        assert te.w.data_ptr() != st.w.data_ptr()
        assert te.nbt.dtype == torch.float32 == ref_t["nbt"].dtype
        for k in ("w", "rm", "nbt"):
            assert rel(getattr(te, k), ref_t[k]) < 1e-6, (it, k)


@pytest.mark.parametrize("shape", [(3, 8, 1000), (2, 5, 37), (4, 64, 125), (2, 3, 12), (5, 6, 64)])
def test_fused_bn_relu_maxpool(shape, dev):
    """Stem fusion: maxpool(relu(bn(x))) forward and its backward vs the unfused torch graph (train-mode BN)."""
    N, C, L = shape
    x = (rnd(1, N, C, L) * 1.3 + 0.2).requires_grad_(True)
    g = (1.0 + 0.2 * rnd(2, C)).requires_grad_(True); b = (0.1 * rnd(3, C)).requires_grad_(True)
    y_ref = F.max_pool1d(F.relu(F.batch_norm(x, None, None, g, b, training=True, eps=1e-5)), 3, 2, 1)
    dy = rnd(4, *y_ref.shape)
    gx, gg, gb = torch.autograd.grad(y_ref, (x, g, b), dy)
    xg = x.detach().to(dev)
    partial = torch.stack([xg.double().sum(dim=(0, 2)).float(), (xg.double() ** 2).sum(dim=(0, 2)).float()], dim=1)[None].contiguous()
    mean, invstd = ops.bn_stats_finalize(partial, N * L, 1e-5, 0.1)
    gd, bd = g.detach().to(dev), b.detach().to(dev)
    y = ops.bn_relu_maxpool_fwd(xg, mean, invstd, gd, bd)
    assert rel(y, y_ref) < 1e-5
    part = ops.bn_relu_maxpool_bwd_reduce(dy.to(dev), xg, mean, invstd, gd, bd)
    sums, dgam, dbet = ops.bn_reduce_partials(part, want_param_grads=True)
    dx = ops.bn_relu_maxpool_bwd_apply(dy.to(dev), xg, mean, invstd, gd, bd, sums, N * L)
    assert rel(dx, gx) < 3e-5 and rel(dgam, gg) < 3e-5 and rel(dbet, gb) < 3e-5
    # eval mode (folded scale / shift)
    rm, rv = 0.2 * rnd(5, C), 1.0 + 0.5 * rnd(6, C).abs()
    sc, sh = ops.bn_fold(gd, bd, rm.to(dev), rv.to(dev), 1e-5)
    ye = ops.bn_relu_maxpool_fwd(xg, None, None, sc, sh)
    ye_ref = F.max_pool1d(F.relu(F.batch_norm(x.detach(), rm, rv, g.detach(), b.detach(), training=False, eps=1e-5)), 3, 2, 1)
    assert rel(ye, ye_ref) < 1e-5


@pytest.mark.parametrize("shape", [(3, 8, 1000), (2, 5, 37)])
def test_fused_bn_relu_maxpool_routes_on_rounded_activations(shape, dev):
    """lp (use_amp, 16-bit stem): the pooled gradient goes to the first maximum of the bf16-ROUNDED activation, as when the
    pool reads BatchNorm's bf16 output under autocast; ties between rounded neighbours are what the flag is about, so the
    test also checks that the input has them (the unrounded routing lands somewhere else)."""
    N, C, L = shape
    x = (rnd(11, N, C, L) * 1.3 + 0.2).requires_grad_(True)
    g = (1.0 + 0.2 * rnd(12, C)).requires_grad_(True); b = (0.1 * rnd(13, C)).requires_grad_(True)
    a = F.relu(F.batch_norm(x, None, None, g, b, training=True, eps=1e-5))
    a_r = a + (a.bfloat16().float() - a).detach()           # rounded values, straight-through gradient
    y_ref = F.max_pool1d(a_r, 3, 2, 1)
    dy = rnd(14, *y_ref.shape)
    gx, gg, gb = torch.autograd.grad(y_ref, (x, g, b), dy)
    xg = x.detach().to(dev)
    partial = torch.stack([xg.double().sum(dim=(0, 2)).float(), (xg.double() ** 2).sum(dim=(0, 2)).float()], dim=1)[None].contiguous()
    mean, invstd = ops.bn_stats_finalize(partial, N * L, 1e-5, 0.1)
    gd, bd = g.detach().to(dev), b.detach().to(dev)
    out = {}
    for lp in (True, False):
        part = ops.bn_relu_maxpool_bwd_reduce(dy.to(dev), xg, mean, invstd, gd, bd, lp=lp)
        sums, dgam, dbet = ops.bn_reduce_partials(part, want_param_grads=True)
        out[lp] = (ops.bn_relu_maxpool_bwd_apply(dy.to(dev), xg, mean, invstd, gd, bd, sums, N * L, lp=lp), dgam, dbet)
    dx, dgam, dbet = out[True]
    # a rounding that lands on the other side of a bf16 boundary on the two devices moves one window: 1e-3, not 3e-5
    assert rel(dx, gx) < 1e-3 and rel(dgam, gg) < 1e-3 and rel(dbet, gb) < 1e-3
    if N * C * L > 10000:        # enough windows for some to hold two equal rounded maxima
        assert rel(out[False][0], gx) > 10 * max(rel(dx, gx), 1e-4)


@pytest.mark.parametrize("case", [(3, 64, 500, 64, 3, 1, 1), (5, 128, 63, 256, 3, 1, 1), (4, 256, 37, 128, 1, 1, 0), (2, 64, 41, 48, 3, 2, 1),
                                  # Winograd forward AND weight gradient (both channel counts multiples of 128), odd / tiny lengths
                                  (3, 128, 125, 128, 3, 1, 1), (2, 512, 63, 128, 3, 1, 1), (130, 128, 5, 256, 3, 1, 1), (1024, 128, 250, 128, 3, 1, 1)])
def test_conv_with_fused_input_bn_relu(case, dev, wino, monkeypatch):
    """conv1d_fwd / conv1d_wgrad with the producer's BN + ReLU applied in the gather == conv on the materialised
    activation (zero padding applied AFTER the activation)."""
    N, Cin, Lin, Cout, K, s, p = case
    c = rnd(1, N, Cin, Lin) * 1.5
    A = 1.0 + 0.3 * rnd(2, Cin); B = 0.4 * rnd(3, Cin)
    a = F.relu(c * A[None, :, None] + B[None, :, None]).requires_grad_(True)
    w = rnd(4, Cout, Cin, K, std=0.1).requires_grad_(True)
    y_ref = F.conv1d(a, w, stride=s, padding=p)
    dy = rnd(5, *y_ref.shape)
    (dw_ref,) = torch.autograd.grad(y_ref, (w,), dy)
    aff = (A.to(dev), B.to(dev))
    y, stats = ops.conv1d_fwd(c.to(dev), w.detach().to(dev), s, p, 1, want_stats=True, in_affine=aff)
    assert rel(y, y_ref) < 2e-5
    sums = ops.bn_reduce_partials(stats).cpu()
    assert rel(sums[:, 1], (y_ref.detach().double() ** 2).sum(dim=(0, 2))) < 2e-5
    dw = ops.conv1d_wgrad(dy.to(dev), c.to(dev), K, s, p, 1, x_affine=aff)       # Winograd shapes: transpose of F(4,3)
    assert rel(dw, dw_ref) < 2e-5
    monkeypatch.setattr(ops, "WINO_WGRAD_F", 2)                                   # ... of F(2,3)
    assert rel(ops.conv1d_wgrad(dy.to(dev), c.to(dev), K, s, p, 1, x_affine=aff), dw_ref) < 2e-5


def test_bn_finalize_affine_outputs(dev):
    C, N, L = 48, 3, 50
    x = rnd(1, N, C, L).to(dev)
    g, b = (1 + 0.2 * rnd(2, C)).to(dev), (0.1 * rnd(3, C)).to(dev)
    partial = torch.stack([x.double().sum(dim=(0, 2)).float(), (x.double() ** 2).sum(dim=(0, 2)).float()], dim=1)[None].contiguous()
    mean, invstd, (A, B) = ops.bn_stats_finalize(partial, N * L, 1e-5, 0.1, affine_of=(g, b))
    y = ops.bn_apply_fwd(x, mean, invstd, g, b, None, True)
    assert torch.equal(torch.relu(torch.addcmul(B[None, :, None], x, A[None, :, None])), y) or rel(torch.relu(x * A[None, :, None] + B[None, :, None]), y) < 1e-6
    sums = ops.bn_reduce_partials(partial)
    m2, i2, (A2, B2) = ops.bn_finalize(sums, N * L, 1e-5, 0.1, affine_of=(g, b))
    assert torch.equal(A, A2) and torch.equal(B, B2) and torch.equal(mean, m2)


@pytest.mark.parametrize("N,K,L", [(1, 4, 1), (3, 4, 2000), (5, 7, 333), (2, 32, 4097)])
def test_seg_confusion_and_mean_iou(N, K, L, dev):
    """Per-record confusion counts (bit-exact integers) and the MeanIoU built on them vs the numpy restatement of
    torchmetrics 1.5.2 (oracle/metrics_ref.py)."""
    from oracle import metrics_ref as M
    from utils.perf_metrics import build_metric_fn
    rng = np.random.RandomState(N * 100 + K)
    pred = rng.randint(0, max(K - 1, 1), size=(N, L))     # the last class is never predicted ...
    target = rng.randint(0, K, size=(N, L))
    target[0] = pred[0]                                   # ... one perfect record
    if N > 1:
        target[1] = 0                                     # one single-class record (empty unions elsewhere)
    counts = SF.seg_confusion(torch.from_numpy(pred).to(dev), torch.from_numpy(target).to(dev), K).cpu().numpy()
    ref = np.zeros((N, K, K), dtype=np.int64)
    for n in range(N):
        np.add.at(ref[n], (target[n], pred[n]), 1)
    assert counts.dtype == np.int32 and np.array_equal(counts, ref)
    for bg in (True, False):
        for per_class in (False, True):
            fn, best = build_metric_fn({"task": "segmentation", "num_classes": K, "include_background": bg, "per_class": per_class,
                                        "input_format": "index", "target_metrics": ["MeanIoU"]})
            assert best == {"MeanIoU": -float("inf")}
            o = M.MeanIoURef(K, bg, per_class)
            for lo, hi in ((0, max(N // 2, 1)), (max(N // 2, 1), N)):      # two (possibly ragged) batches
                if hi > lo:
                    fn.update(torch.from_numpy(pred[lo:hi]).to(dev), torch.from_numpy(target[lo:hi]).to(dev))
                    o.update(pred[lo:hi], target[lo:hi])
            got = fn.compute()["MeanIoU"].cpu().numpy()
            assert np.allclose(got, o.compute(), rtol=0, atol=1e-6), (bg, per_class)
    # out-of-range labels are skipped, one-hot inputs are accepted
    t2 = torch.from_numpy(target).to(dev).clone(); t2[:, 0] = -1
    c2 = SF.seg_confusion(torch.from_numpy(pred).to(dev), t2, K).cpu().numpy()
    assert np.array_equal(c2.sum(axis=(1, 2)), np.full(N, L - 1))
    fn, _ = build_metric_fn({"task": "segmentation", "num_classes": K, "target_metrics": ["MeanIoU"]})
    oh = lambda a: torch.nn.functional.one_hot(torch.from_numpy(a), K).movedim(-1, 1).to(dev)
    fn.update(oh(pred), oh(target))
    o = M.MeanIoURef(K); o.update(pred, target)
    assert abs(float(fn.compute()["MeanIoU"]) - o.compute()) < 1e-6
