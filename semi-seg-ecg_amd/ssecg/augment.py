"""On-device strong augmentation + standardisation of the unlabelled batch.

``DeviceStrongAugment`` is built from the reference's own YAML block (``dataset.strong_augmentations`` with one
``RandAugment`` over AmplitudeScaling / AdaptivePowerlineNoise / RandomPartialWhiteNoise / RandomPartialSineNoise,
configs/base/resnet18/fixmatch.yaml:62-77) and turns a raw weak-view batch (B, C, L) on the device into the two
tensors the step consumes: ``ecg`` = standardize(x) and ``ecg_aug`` = standardize(RandAugment(x))
(src/utils/semi_dataset.py:235-244).  The host only draws the per-record plan (a few integers per record, from the
repo's counter-based generator - reproducible for a (seed, step) pair on any rank); the signal never leaves HBM."""
from __future__ import annotations

import numpy as np
import torch

from . import ops, synth

OP_IDS = {"AmplitudeScaling": 0, "amplitude_scaling": 0, "AdaptivePowerlineNoise": 1, "adaptive_powerline_noise": 1,
          "RandomPartialWhiteNoise": 2, "partial_white_noise": 2, "RandomPartialSineNoise": 3, "partial_sine_noise": 3}
PLAN_W = 12


def make_plans(uniforms: np.ndarray, L: int, op_ids=(0, 1, 2, 3), num_layers: int = 3, prob: float = 0.5,
               ratio_white: float = 0.5, ratio_sine: float = 0.5) -> np.ndarray:
    """(B, 16) U(0,1) draws -> (B, 12) int32 plans: ``num_layers`` ops without replacement (partial Fisher-Yates, the law
    of np.random.choice(.., replace=False)), each fired with probability ``prob``, 50/60 Hz with probability 1/2,
    count = int(U(0, ratio) * L), start = randint(0, L - count)   (src/utils/transforms.py:494-499, 536-541, 574-583, 647-649)."""
    u = np.asarray(uniforms, dtype=np.float64)
    B, n = u.shape[0], len(op_ids)
    if num_layers > n or num_layers > 4:
        raise ValueError("RandAugment: num_layers exceeds the number of ops (np.random.choice(replace=False) would raise)")
    plans = np.zeros((B, PLAN_W), dtype=np.int32)
    perm = np.tile(np.asarray(op_ids, dtype=np.int32), (B, 1))
    rows = np.arange(B)
    for k in range(num_layers):
        j = k + (u[:, k] * (n - k)).astype(np.int64)
        pk, pj = perm[rows, k].copy(), perm[rows, j].copy()
        perm[rows, k], perm[rows, j] = pj, pk
    plans[:, :num_layers] = perm[:, :num_layers]
    for k in range(num_layers):
        plans[:, 4] |= (u[:, 4 + k] < prob).astype(np.int32) << k
    plans[:, 5] = np.where(u[:, 8] < 0.5, 50, 60)
    for col, uc, us, ratio in ((6, 9, 10, ratio_white), (8, 11, 12, ratio_sine)):
        count = (u[:, uc] * ratio * L).astype(np.int64)
        plans[:, col] = count
        plans[:, col + 1] = (u[:, us] * (L - count)).astype(np.int64)
    plans[:, 10] = num_layers
    return plans


class DeviceStrongAugment:
    def __init__(self, strong_augmentations, seed: int = 0):
        if not (isinstance(strong_augmentations, (list, tuple)) and len(strong_augmentations) == 1
                and isinstance(strong_augmentations[0], dict) and "RandAugment" in strong_augmentations[0]):
            raise NotImplementedError("the device pipeline covers `strong_augmentations: [RandAugment: {...}]`")
        ra = strong_augmentations[0]["RandAugment"]
        level = ra.get("level", 10) / 10.0                      # _set_level(level, max_level=10)
        self.num_layers = int(ra.get("num_layers", 2))
        self.prob = float(ra.get("prob", 0.5))
        self.sigma = level * 0.5                                # AmplitudeScaling._set_level  (transforms.py:350-351)
        self.amplitude = level * 1.0                            # _Noise._set_level            (:452-455)
        self.sine_freq = 0.5 / level
        self.ratio = level * 0.5                                # _RandomPartialNoise._set_level (:548-550)
        self.fs = 500.0                                         # AdaptivePowerlineNoise default (:485)
        self.op_ids = []
        for op in ra["ops"]:
            name, kwargs = (op, {}) if isinstance(op, str) else list(op.items())[0]
            if name not in OP_IDS:
                raise NotImplementedError(f"strong augmentation {name!r} stays on the reference's host pipeline")
            self.op_ids.append(OP_IDS[name])
            if OP_IDS[name] == 1:
                self.fs = float((kwargs or {}).get("fs", 500))
        if len(set(self.op_ids)) != len(self.op_ids):
            raise NotImplementedError("each op may appear once in RandAugment.ops")
        self.seed = int(seed)
        self.calls = 0

    def plans(self, B: int, L: int, step: int) -> np.ndarray:
        u = synth.uniform(self.seed, 40, B * 16, offset=step * B * 16).reshape(B, 16)
        return make_plans(u, L, self.op_ids, self.num_layers, self.prob, self.ratio, self.ratio)

    @torch.no_grad()
    def __call__(self, x: torch.Tensor, step: int | None = None, plans=None, scales=None, white=None):
        """x: raw weak-view batch (B, C, L) fp32 on the device -> (ecg, ecg_aug), both standardised fp32."""
        if step is None:
            step, self.calls = self.calls, self.calls + 1
        B, _, L = x.shape
        if plans is None:
            plans = self.plans(B, L, step)
        plan_dev = torch.from_numpy(np.ascontiguousarray(plans, dtype=np.int32)).pin_memory().to(x.device, non_blocking=True)
        aug = ops.strong_augment(x, plan_dev, self.sigma, self.fs, self.amplitude, self.sine_freq,
                                 seed=self.seed * 1000003 + step, scales=scales, white=white)
        return ops.standardize(x), ops.standardize(aug, out=aug)


_ACTIVE = [None]


def configure(dataset_cfg: dict, seed: int = 0):
    """Called by the plugins' ``train(config)``: with ``dataset.device_augment: true`` the unlabelled loader hands over
    the raw weak view (key ``ecg_raw``) and both standardised views are produced on the device from the YAML's own
    ``strong_augmentations`` block.  Returns the active augmenter (None = the host pipeline delivers ``ecg_aug``)."""
    _ACTIVE[0] = None
    if dataset_cfg.get("device_augment"):
        _ACTIVE[0] = DeviceStrongAugment(dataset_cfg["strong_augmentations"], seed=seed)
    return _ACTIVE[0]


def unlabeled_views(batch: dict, device, want_strong: bool = True):
    """-> (ecg_u_w, ecg_u_s or None) on the device, from either loader convention: the reference's
    ``{'ecg', 'ecg_aug'}`` (host pipeline, passed through) or ``{'ecg_raw'}`` (device pipeline)."""
    if "ecg_raw" not in batch:
        weak = batch["ecg"].to(device, non_blocking=True)
        return weak, (batch["ecg_aug"].to(device, non_blocking=True) if want_strong else None)
    raw = batch["ecg_raw"].to(device, non_blocking=True)
    if not want_strong:
        return ops.standardize(raw), None
    if _ACTIVE[0] is None:
        raise RuntimeError("the loader delivers 'ecg_raw' but dataset.device_augment is not configured")
    return _ACTIVE[0](raw)
