"""Dedicated stem kernels (csrc/stem.hip: Conv1d(C -> 64, k 7, stride 2, pad 3)) through the C ABI vs torch-CPU, fp32.

Same bars as tests/test_ops_gpu.py: 2e-5 of the tensor's scale for conv outputs / gradients, BN partial sums against fp64.
Shapes: the reference's lead counts (1, 2, 12), the maximum (16), odd / tiny / long lengths, more tiles than workgroup
slots (persistent loop), tiles ending inside and exactly at a sample's end."""
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


STEM_CASES = [  # N, C, L
    (2, 12, 2000), (3, 1, 2000), (2, 2, 2500), (5, 12, 333), (4, 16, 37), (3, 3, 7), (2, 5, 1), (1, 12, 1024), (2, 12, 511),
    (2, 12, 513), (600, 2, 515), (140, 12, 2000),
]


@pytest.mark.parametrize("case", STEM_CASES)
def test_stem_forward_and_statistics(case, dev):
    N, C, L = case
    x, w = rnd(1, N, C, L), rnd(2, 64, C, 7, std=0.2)
    assert ops._stem_ok(N, C, L, 64, 7, 2, 3, 1)
    y, partial = ops.conv1d_fwd(x.to(dev), w.to(dev), 2, 3, 1, want_stats=True)
    ref = F.conv1d(x.double(), w.double(), stride=2, padding=3)
    assert tuple(y.shape) == tuple(ref.shape)
    assert rel(y, ref) < 2e-5
    s = partial.double().sum(0).cpu()
    assert torch.allclose(s[:, 0], ref.sum((0, 2)), rtol=1e-4, atol=1e-3 * ref.abs().max().item())
    assert torch.allclose(s[:, 1], (ref * ref).sum((0, 2)), rtol=1e-4, atol=1e-6)
    # the generic implicit GEMM is the second implementation of the same contract
    ops.STEM = False
    try:
        y2, _ = ops.conv1d_fwd(x.to(dev), w.to(dev), 2, 3, 1)
    finally:
        ops.STEM = True
    assert rel(y, y2) < 2e-5


@pytest.mark.parametrize("case", STEM_CASES)
def test_stem_eval_fused_pool(case, dev):
    N, C, L = case
    x, w = rnd(3, N, C, L), rnd(4, 64, C, 7, std=0.2)
    scale, shift = rnd(5, 64).abs() + 0.5, rnd(6, 64, std=0.3)
    y = ops.stem_fwd_eval_pool(x.to(dev), w.to(dev), scale.to(dev), shift.to(dev))
    c = F.conv1d(x.double(), w.double(), stride=2, padding=3)
    ref = F.max_pool1d(F.relu(c * scale.double()[None, :, None] + shift.double()[None, :, None]), 3, 2, 1)
    assert tuple(y.shape) == tuple(ref.shape)
    assert rel(y, ref) < 2e-5


def test_stem_eval_fused_pool_propagates_nan(dev):
    """A NaN in the input must reach the pooled output of the fused eval stem exactly where nn.MaxPool1d(ReLU(BN(conv)))
    puts it (a diverged model must not look healthy in the teacher / evaluate passes): ReLU and the pooling maximum keep NaN."""
    N, C, L = 3, 12, 2000
    x, w = rnd(3, N, C, L), rnd(4, 64, C, 7, std=0.2)
    x[1, 5, 777] = float("nan"); x[2, 0, 0] = float("nan"); x[0, 11, L - 1] = float("nan")
    scale, shift = rnd(5, 64).abs() + 0.5, rnd(6, 64, std=0.3)
    y = ops.stem_fwd_eval_pool(x.to(dev), w.to(dev), scale.to(dev), shift.to(dev)).cpu()
    c = F.conv1d(x, w, stride=2, padding=3)
    ref = F.max_pool1d(F.relu(c * scale[None, :, None] + shift[None, :, None]), 3, 2, 1)
    assert torch.equal(torch.isnan(y), torch.isnan(ref)) and torch.isnan(ref).sum() > 0
    ok = ~torch.isnan(ref)
    assert ((y[ok] - ref[ok]).abs().max() / ref[ok].abs().max()).item() < 2e-5


@pytest.mark.parametrize("case", STEM_CASES)
def test_stem_weight_gradient(case, dev):
    N, C, L = case
    x = rnd(7, N, C, L)
    Lout = (L - 1) // 2 + 1
    dc = rnd(8, N, 64, Lout)
    dw = ops.conv1d_wgrad(dc.to(dev), x.to(dev), 7, 2, 3, 1)
    w = torch.zeros(64, C, 7, dtype=torch.float64, requires_grad=True)
    F.conv1d(x.double(), w, stride=2, padding=3).backward(dc.double())
    assert tuple(dw.shape) == (64, C, 7)
    assert rel(dw, w.grad) < 2e-5
    dw2 = ops.conv1d_wgrad(dc.to(dev), x.to(dev), 7, 2, 3, 1)
    assert torch.equal(dw, dw2)   # fixed-order slab sums: bitwise reproducible


def test_stem_node_train_and_eval_match_unfused_chain(dev):
    """StemFn end to end (train: conv + stats -> BN + ReLU + pool, backward; eval: the single fused launch) with the
    dedicated kernels on and off."""
    N, C, L = 4, 12, 2000
    x = rnd(11, N, C, L).to(dev)
    outs = {}
    for stem in (True, False):
        ops.STEM = stem
        try:
            w = rnd(12, 64, C, 7, std=0.2).to(dev).requires_grad_(True)
            g = (rnd(13, 64).abs() + 0.5).to(dev).requires_grad_(True)
            b = rnd(14, 64, std=0.1).to(dev).requires_grad_(True)
            bn = SF.BNState(g, b, torch.zeros(64, device=dev), torch.ones(64, device=dev), None, 1e-5, 0.1, None)
            y = SF.StemFn.apply(x, w, g, b, bn, True)
            (y * rnd(15, *y.shape).to(dev)).sum().backward()
            bn_e = SF.BNState(g.detach(), b.detach(), bn.running_mean.clone(), bn.running_var.clone(), None, 1e-5, 0.1, None)
            with torch.no_grad():
                ye = SF.StemFn.apply(x, w.detach(), g.detach(), b.detach(), bn_e, False)
            outs[stem] = (y.detach(), w.grad, g.grad, b.grad, ye)
        finally:
            ops.STEM = True
    for a, b_ in zip(outs[True], outs[False]):
        assert rel(a, b_) < 5e-5


@pytest.mark.gpu
@pytest.mark.parametrize("Na,Nb,C,L", [(5, 7, 12, 2000), (3, 2, 1, 250), (1, 9, 2, 333), (16, 16, 12, 500)])
def test_two_source_stem_equals_the_concatenated_batch_bit_for_bit(Na, Nb, C, L):
    """ssecg_stem_fwd2 / ssecg_stem_wgrad2 (round 4): the student batch of the semi-supervised plugins as two tensors
    (labelled, unlabelled) instead of torch.cat((a, b)) - conv output, BN partial sums and the weight gradient must equal the
    one-source kernels on the concatenation bit for bit (aligned 16-byte staging at L % 4 == 0 and the dword path otherwise)."""
    from ssecg import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(Na * 100 + Nb)
    a = torch.randn(Na, C, L, generator=g).to(dev); b = torch.randn(Nb, C, L, generator=g).to(dev)
    w = (torch.randn(64, C, 7, generator=g) * 0.1).to(dev)
    pair = ops.BatchPair(a, b)
    assert ops.stem_pair_ok(pair, w) and pair.size() == torch.Size((Na + Nb, C, L)) and pair.size(2) == L
    c2, s2 = ops.stem_fwd_pair(pair, w)
    c1, s1 = ops.conv1d_fwd(pair.cat(), w, 2, 3, 1, want_stats=True)
    assert torch.equal(c2, c1) and torch.equal(s2, s1)
    dc = torch.randn(c1.shape, generator=torch.Generator().manual_seed(3)).to(dev)
    assert torch.equal(ops.stem_wgrad_pair(dc, pair), ops.conv1d_wgrad(dc, pair.cat(), 7, 2, 3, 1))


@pytest.mark.gpu
def test_model_step_with_a_batch_pair_equals_the_concatenated_batch(monkeypatch):
    """The plugins hand the student batch over as ops.batch_pair(labelled, unlabelled): logits and all 65 gradients equal those of
    model(torch.cat(...)) bit for bit; SSECG_STEM_PAIR=0 (ops.STEM_PAIR = False) makes batch_pair concatenate."""
    from helpers import build_hip_model, dropout_mask_np
    from ssecg import ops, synth
    dev = torch.device("cuda:0")
    C, B, L, seed = 2, 4, 500, 17
    sd_np = synth.model_state(seed, C, trained=True)
    bt = synth.fixmatch_batch(seed + 1, B, C, L)
    xa = torch.from_numpy(bt["labeled"]["ecg"]).to(dev); xb = torch.from_numpy(bt["unlabeled"]["ecg_aug"]).to(dev)
    tl = torch.from_numpy(bt["labeled"]["target"]).to(dev)
    t = torch.cat((tl, tl))
    dm = torch.from_numpy(dropout_mask_np(seed, 2 * B, lp=16)).to(dev, torch.uint8)
    outs = []
    for use_pair in (True, False):
        monkeypatch.setattr(ops, "STEM_PAIR", use_pair)
        inp = ops.batch_pair(xa, xb)
        assert isinstance(inp, ops.BatchPair) == use_pair
        model = build_hip_model(C, sd_np, dev).train()
        model.decode_head.fixed_dropout_mask = dm
        logits = model(inp, return_loss=False)["seg_logits"]
        torch.nn.functional.cross_entropy(logits, t).backward()
        outs.append((logits.detach().clone(), {k: p.grad.detach().clone() for k, p in model.named_parameters()}))
    assert torch.equal(outs[0][0], outs[1][0])
    for k in outs[0][1]:
        assert torch.equal(outs[0][1][k], outs[1][1][k]), k
