"""GPU: the HIP record pipeline (ssecg_strong_augment + ssecg_standardize) against the reference outputs frozen in
tests/golden/augment_*.npz and against the oracle, through the C ABI."""
import numpy as np
import pytest
import torch

from helpers import golden
from oracle import augment_ref as A
from ssecg import augment as P
from ssecg import ops, synth
from ssecg.lib import SsecgError
from test_augment_cpu import raw_records

pytestmark = pytest.mark.gpu
PARAMS = A.level_params(10)


def _dev(a, dev, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return (t if dtype is None else t.to(dtype)).to(dev)


@pytest.mark.parametrize("name", ["augment_c1", "augment_c12", "augment_short"])
def test_views_match_reference(name, dev):
    g = golden(name)
    C, B, Lr, seed, fs = (int(v) for v in g["meta"])
    x = raw_records(seed, B, C, Lr)
    xd = _dev(x, dev)
    raw = ops.strong_augment(xd, _dev(g["plans"], dev), PARAMS["sigma"], fs, PARAMS["amplitude"], PARAMS["sine_freq"],
                             scales=_dev(g["scales"], dev), white=_dev(g["white"], dev))
    o_raw = np.stack([A.strong_augment(x[b], g["plans"][b], g["scales"][b], g["white"][b], fs, PARAMS) for b in range(B)])
    assert np.abs(raw.cpu().numpy() - o_raw).max() <= 1e-6 * np.abs(o_raw).max()      # fp64 in the kernel, fp32 store
    ecg = ops.standardize(xd).cpu().numpy()
    aug = ops.standardize(raw).cpu().numpy()
    assert np.abs(ecg - g["ecg"]).max() < 5e-7                                        # tolerance: 1e-6 absolute on O(1) data
    assert np.abs(aug - g["ecg_aug"]).max() < 3e-6
    # the host class: same numbers through DeviceStrongAugment with the reference's draws injected
    cls = P.DeviceStrongAugment([{"RandAugment": {"ops": [{"AmplitudeScaling": {"sigma": 0.5}}, {"AdaptivePowerlineNoise": {"fs": fs}},
                                                          {"RandomPartialWhiteNoise": {"amplitude": 1, "ratio": 0.5}},
                                                          {"RandomPartialSineNoise": {"amplitude": 1, "ratio": 0.5}}],
                                                  "level": 10, "num_layers": 3, "prob": 0.5}}])
    e2, a2 = cls(xd, step=0, plans=g["plans"], scales=_dev(g["scales"], dev), white=_dev(g["white"], dev))
    assert np.array_equal(e2.cpu().numpy(), ecg) and np.array_equal(a2.cpu().numpy(), aug)


@pytest.mark.parametrize("B,C,L", [(3, 1, 1), (2, 3, 5), (4, 2, 2000), (2, 2, 4096)])
def test_generated_noise_matches_the_counter_generator(B, C, L, dev):
    """scales / white == NULL: the kernel draws them itself; the oracle fed with ssecg.synth's numbers must agree."""
    seed = 1234 + L
    x = raw_records(7, B, C, L)
    u = synth.uniform(seed, 40, B * 16).reshape(B, 16)
    plans = P.make_plans(u, L)
    plans[:, 4] = 0b111                                                   # fire every layer
    if B > 1:
        plans[1, :3] = (0, 2, 1)                                          # scale -> white -> powerline on the result
    got = ops.strong_augment(_dev(x, dev), _dev(plans, dev), 0.5, 250, 1.0, 0.5, seed=seed).cpu().numpy()
    scales = 1.0 + 0.5 * synth.normal(seed, 1, (B, C, L)).astype(np.float64)
    white = synth.normal(seed, 2, (B, C, L)).astype(np.float64)
    ref = np.stack([A.strong_augment(x[b], plans[b], scales[b], white[b], 250, PARAMS) for b in range(B)])
    assert np.abs(got - ref).max() <= 2e-6 * max(np.abs(ref).max(), 1.0)   # synth.normal rounds its draws to fp32
    again = ops.strong_augment(_dev(x, dev), _dev(plans, dev), 0.5, 250, 1.0, 0.5, seed=seed).cpu().numpy()
    other = ops.strong_augment(_dev(x, dev), _dev(plans, dev), 0.5, 250, 1.0, 0.5, seed=seed + 1).cpu().numpy()
    assert np.array_equal(got, again) and not np.array_equal(got, other)


def test_edge_cases(dev):
    L = 64
    x = np.zeros((3, 2, L), dtype=np.float32)
    x[0] = 2.5                                                            # flat record -> zeros after standardisation
    x[1] = np.linspace(-1, 1, 2 * L, dtype=np.float32).reshape(2, L)
    x[2, 0, 5] = 1.0
    y = ops.standardize(_dev(x, dev)).cpu().numpy()
    assert np.array_equal(y[0], np.zeros((2, L), dtype=np.float32))
    assert np.abs(y - A.standardize(x)).max() < 5e-7
    xd = _dev(x, dev)
    assert ops.standardize(xd, out=xd) is xd and np.array_equal(xd.cpu().numpy(), y)          # in place
    plans = np.zeros((3, 12), dtype=np.int32)
    plans[:, :3] = (2, 3, 1); plans[:, 10] = 3
    plans[0, 4] = 0                                                       # nothing fires -> copy
    plans[1, 4] = 0b011; plans[1, 6:10] = (0, L - 1, L // 2, L // 2)      # count 0; sine noise up to the last sample
    plans[2, 4] = 0b100; plans[2, 5] = 60                                 # powerline on a spike: p95 == p5 == 0 -> no change
    w = synth.normal(1, 2, (3, 2, L))
    got = ops.strong_augment(_dev(x, dev), _dev(plans, dev), 0.5, 250, 1.0, 0.5, white=_dev(w, dev),
                             scales=_dev(np.ones_like(x), dev)).cpu().numpy()
    ref = np.stack([A.strong_augment(x[b], plans[b], np.ones((2, L)), w[b], 250, PARAMS) for b in range(3)])
    assert np.array_equal(got[0], x[0]) and np.array_equal(got[2], x[2])
    assert np.abs(got - ref).max() < 1e-6
    with pytest.raises(SsecgError):
        ops.strong_augment(_dev(np.zeros((1, 1, 4097), dtype=np.float32), dev), _dev(plans[:1], dev), 0.5, 250, 1.0, 0.5)
    with pytest.raises(SsecgError):
        ops.strong_augment(_dev(x, dev), _dev(plans[:2], dev), 0.5, 250, 1.0, 0.5)
    with pytest.raises(SsecgError):
        ops.standardize(torch.zeros(2, 3))


def test_full_size_properties(dev):
    """B = 512, C = 12, L = 2000 (the bench batch): standardisation yields zero mean / unit population std per record and
    is idempotent; records whose plan fires nothing come out identical in both views; the strong view never leaves HBM."""
    B, C, L = 512, 12, 2000
    x = torch.from_numpy((0.2 + 1.5 * synth.normal(5, 1, (B, C, L))).astype(np.float32)).to(dev)
    cls = P.DeviceStrongAugment([{"RandAugment": {"ops": ["AmplitudeScaling", {"AdaptivePowerlineNoise": {"fs": 250}},
                                                          "RandomPartialWhiteNoise", "RandomPartialSineNoise"],
                                                  "level": 10, "num_layers": 3, "prob": 0.5}}], seed=11)
    plans = cls.plans(B, L, step=0)
    ecg, aug = cls(x, step=0, plans=plans)
    for t in (ecg, aug):
        m = t.double().mean(dim=(1, 2)); s = t.double().std(dim=(1, 2), unbiased=False)
        assert m.abs().max().item() < 1e-6 and (s - 1).abs().max().item() < 1e-5
    again = ops.standardize(ecg)
    assert (again - ecg).abs().max().item() < 2e-6                      # idempotent
    none = torch.from_numpy(plans[:, 4] == 0).to(dev)
    assert none.any() and torch.equal(aug[none], ecg[none])              # nothing fired -> the two views coincide
    fired = ~none
    assert (aug[fired] - ecg[fired]).abs().amax(dim=(1, 2)).min().item() > 1e-3
    assert ecg.is_cuda and aug.is_cuda and aug.dtype == torch.float32
