#!/usr/bin/env python3
"""Time the on-device record pipeline (ssecg_strong_augment + 2x ssecg_standardize) at the bench shape and put it
beside the numpy restatement of the reference's per-record host pipeline (oracle/augment_ref.py, one core).
Usage: python tools/augment_bench.py [B C L]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semi-seg-ecg_amd"))
sys.path.insert(0, ROOT)
from ssecg import augment as P, ops, synth  # noqa: E402

B, C, L = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (512, 12, 2000)
dev = torch.device("cuda:0")
x = torch.from_numpy((0.2 + 1.5 * synth.normal(1, 1, (B, C, L))).astype(np.float32)).to(dev)
aug = P.DeviceStrongAugment([{"RandAugment": {"ops": ["AmplitudeScaling", {"AdaptivePowerlineNoise": {"fs": 250}},
                                                      "RandomPartialWhiteNoise", "RandomPartialSineNoise"],
                                              "level": 10, "num_layers": 3, "prob": 0.5}}], seed=3)
for _ in range(3):
    aug(x)
torch.cuda.synchronize()
n = 20
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(n):
    aug(x, step=i)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
# kernel-only: plans resident
plan = torch.from_numpy(aug.plans(B, L, 0)).to(dev)
e0.record()
for i in range(n):
    raw = ops.strong_augment(x, plan, aug.sigma, aug.fs, aug.amplitude, aug.sine_freq, seed=i)
    ops.standardize(x); ops.standardize(raw, out=raw)
e1.record(); torch.cuda.synchronize()
ms_k = e0.elapsed_time(e1) / n
bytes_alg = B * C * L * 4 * (2 + 2 + 2)   # aug: read x, write raw; standardize x2: read + write each (re-reads hit L2)
print(f"device pipeline B={B} C={C} L={L}: {ms:.3f} ms/batch incl. host plan + H2D, {ms_k:.3f} ms kernels only "
      f"-> {B / ms_k * 1e3:.0f} records/s, {bytes_alg / ms_k / 1e6:.1f} GB/s algorithmic ({bytes_alg / ms_k / 1e6 / 8000:.3f} of 8 TB/s)")
from oracle import augment_ref as A  # noqa: E402  (checker / CPU baseline only)
xs = x[:8].cpu().numpy()
plans = aug.plans(8, L, 0)
sc = 1 + 0.5 * synth.normal(5, 1, (8, C, L)); wh = synth.normal(5, 2, (8, C, L))
t0 = time.perf_counter()
A.weak_and_strong_views(xs, plans, sc, wh, 250, A.level_params(10))
dt = (time.perf_counter() - t0) / 8
print(f"numpy restatement of the host pipeline: {dt * 1e3:.2f} ms/record on one core -> {1 / dt:.0f} records/s/core")
