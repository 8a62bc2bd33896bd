"""TEST INFRASTRUCTURE ONLY - numpy (float64) restatement of the reference's strong-augmentation + standardisation
record pipeline for the unlabelled loader.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this.

Follows src/utils/transforms.py: Standardize :290-310, AmplitudeScaling :340-351, _Noise._set_level :452-455,
AdaptivePowerlineNoise :480-502, SineNoise :504-509, WhiteNoise :518-522, _RandomPartialNoise :524-550,
RandomApply :567-583, RandAugment :628-657, and the call order of src/utils/semi_dataset.py:235-244
(``ecg = transform(x)``, ``ecg_aug = transform(strong_aug_fn(x))`` with transform = standardize -> float32 tensor).

The reference draws from numpy's global MT19937 stream, which a GPU cannot replay; the random DECISIONS are therefore
an explicit input here (the "plan" + the two noise arrays).  tools/make_golden.py records the draws the reference made
and pins this file against the reference's outputs for those same draws (tests/golden/augment_*.npz).
"""
import numpy as np

OP_AMPLITUDE_SCALING, OP_POWERLINE, OP_PARTIAL_WHITE, OP_PARTIAL_SINE = 0, 1, 2, 3
OP_NAMES = {"AmplitudeScaling": 0, "AdaptivePowerlineNoise": 1, "RandomPartialWhiteNoise": 2, "RandomPartialSineNoise": 3}
PLAN_W = 12   # int32 per record: op[0..3], applied bitmask, powerline Hz, white count/start, sine count/start, layers, pad


def level_params(level=10, max_level=10):
    """What RandAugment.__init__ -> op._set_level(level) leaves in the four ops (transforms.py:350-351, 452-455, 548-550):
    sigma = level/max*0.5; noise amplitude = level/max, sine 'freq' = 0.5/(level/max); partial ratio = level/max*0.5."""
    lv = level / max_level
    return {"sigma": lv * 0.5, "amplitude": lv * 1.0, "sine_freq": 0.5 / lv, "ratio": lv * 0.5}


def standardize(x):
    """transforms.py:301-310 with axis=(-1,-2): per record over (leads, time); 0 where std == 0.  x: (..., C, L)."""
    x = np.asarray(x, dtype=np.float64)
    loc = np.mean(x, axis=(-1, -2), keepdims=True)
    scale = np.std(x, axis=(-1, -2), keepdims=True)
    return np.divide(x - loc, scale, out=np.zeros_like(x), where=scale != 0)


def percentile_linear(row, q):
    """np.percentile(row, q) (method='linear') of a 1-D array, restated: virtual index (n-1)*q/100, numpy's _lerp."""
    a = np.sort(np.asarray(row, dtype=np.float64))
    h = (a.size - 1) * (q / 100.0)
    lo = int(np.floor(h))
    hi = min(lo + 1, a.size - 1)
    t = h - lo
    d = a[hi] - a[lo]
    return a[lo] + d * t if t < 0.5 else a[hi] - d * (1.0 - t)


def powerline_noise(x, fs, freq_hz):
    """transforms.py:488-502: per lead amplitude (p95 - p5)/2 of the CURRENT signal, sin(2*pi*f*t), t = arange(L)/fs."""
    C, L = x.shape
    t = np.expand_dims(np.arange(L) / fs, axis=0)
    amp = np.array([[(percentile_linear(x[c], 95) - percentile_linear(x[c], 5)) / 2] for c in range(C)])
    return amp * np.sin(2 * np.pi * freq_hz * t)


def partial(noise, count, start, shape):
    """transforms.py:536-542: noise[:, :count] dropped at [start, start+count)."""
    out = np.zeros(shape)
    out[:, start:start + count] = noise[:, :count]
    return out


def strong_augment(x, plan, scales, white, fs, params):
    """One record.  x: (C, L); plan: PLAN_W ints; scales: (C, L) multiplicative factors N(1, sigma) used by
    AmplitudeScaling; white: (C, L) standard-normal draws used by RandomPartialWhiteNoise; -> (C, L) float64."""
    x = np.asarray(x, dtype=np.float64)
    C, L = x.shape
    for k in range(int(plan[10])):
        if not (int(plan[4]) >> k) & 1:
            continue                                                   # RandomApply did not fire (transforms.py:575)
        op = int(plan[k])
        if op == OP_AMPLITUDE_SCALING:
            x = x * np.asarray(scales, dtype=np.float64)               # :346-348
        elif op == OP_POWERLINE:
            x = x + powerline_noise(x, fs, int(plan[5]))               # :448-450
        elif op == OP_PARTIAL_WHITE:
            noise = params["amplitude"] * np.asarray(white, dtype=np.float64)   # :521-522
            x = x + partial(noise, int(plan[6]), int(plan[7]), x.shape)
        elif op == OP_PARTIAL_SINE:
            t = np.expand_dims(np.arange(L) / L, axis=0)
            noise = params["amplitude"] * np.sin(2 * np.pi * t / params["sine_freq"])   # :507-509
            x = x + partial(np.broadcast_to(noise, x.shape), int(plan[8]), int(plan[9]), x.shape)
        else:
            raise ValueError(f"unknown op id {op}")
    return x


def weak_and_strong_views(x, plans, scales, white, fs, params):
    """Batch (B, C, L) -> (ecg, ecg_aug) float32, as the unlabelled dataset item (semi_dataset.py:235-244)."""
    ecg = standardize(x).astype(np.float32)
    aug = np.stack([strong_augment(x[b], plans[b], scales[b], white[b], fs, params) for b in range(x.shape[0])])
    return ecg, standardize(aug).astype(np.float32)


def make_plans(uniforms, L, num_ops=4, num_layers=3, prob=0.5, ratio=0.5):
    """Plans from i.i.d. U(0,1) draws, uniforms: (B, 16) - the decisions RandAugment/RandomApply/the ops make
    (choice without replacement = partial Fisher-Yates; rand() < prob; 50/60 Hz; count = int(U(0,ratio)*L);
    start = randint(0, L - count)).  Same rule as ssecg.augment.make_plans (the product's plan generator)."""
    B = uniforms.shape[0]
    plans = np.zeros((B, PLAN_W), dtype=np.int32)
    for b in range(B):
        u = uniforms[b]
        ops = list(range(num_ops))
        for k in range(num_layers):
            j = k + int(u[k] * (num_ops - k))
            ops[k], ops[j] = ops[j], ops[k]
        plans[b, :num_layers] = ops[:num_layers]
        plans[b, 4] = sum(1 << k for k in range(num_layers) if u[4 + k] < prob)
        plans[b, 5] = 50 if u[8] < 0.5 else 60
        for col, (uc, us) in ((6, (9, 10)), (8, (11, 12))):
            count = int(u[uc] * ratio * L)
            plans[b, col] = count
            plans[b, col + 1] = int(u[us] * (L - count))
        plans[b, 10] = num_layers
    return plans
