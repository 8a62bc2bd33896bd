#!/usr/bin/env python3
"""fp32 rounding error of the three conv forms used on the hot path - direct, Winograd F(2,3), Winograd F(4,3) - against an
fp64 convolution, on the host (numpy), for He-initialised weights and post-ReLU inputs.  The numbers quoted in
csrc/conv_wino4.hip and DESIGN.md come from here.  usage: python tools/wino_numerics.py"""
import numpy as np


def direct(x, w, dt):
    x, w = x.astype(dt), w.astype(dt)
    L = x.shape[1]
    xp = np.zeros((x.shape[0], L + 2), dt); xp[:, 1:L + 1] = x
    return sum(w[:, :, t] @ xp[:, t:t + L] for t in range(3))


def wino23(x, w):
    f = np.float32
    C, L = x.shape; Lh = (L + 1) // 2
    xp = np.zeros((C, 2 * Lh + 2), f); xp[:, 1:L + 1] = x
    d = [xp[:, i:i + 2 * Lh:2] for i in range(4)]
    v = [d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]]
    g0, g1, g2 = (w[:, :, t].astype(f) for t in range(3))
    u = [g0, (g0 + g1 + g2) * f(0.5), (g0 - g1 + g2) * f(0.5), g2]
    m = [u[k] @ v[k] for k in range(4)]
    y = np.empty((w.shape[0], 2 * Lh), f)
    y[:, 0::2] = m[0] + m[1] + m[2]; y[:, 1::2] = m[1] - m[2] - m[3]
    return y[:, :L]


def wino43(x, w):
    f = np.float32
    C, L = x.shape; Lq = (L + 3) // 4
    xp = np.zeros((C, 4 * Lq + 2), f); xp[:, 1:L + 1] = x
    d = [xp[:, i:i + 4 * Lq:4] for i in range(6)]
    v = [f(4) * d[0] - f(5) * d[2] + d[4], -f(4) * d[1] - f(4) * d[2] + d[3] + d[4], f(4) * d[1] - f(4) * d[2] - d[3] + d[4],
         -f(2) * d[1] - d[2] + f(2) * d[3] + d[4], f(2) * d[1] - d[2] - f(2) * d[3] + d[4], f(4) * d[1] - f(5) * d[3] + d[5]]
    g0, g1, g2 = (w[:, :, t].astype(f) for t in range(3))
    u = [g0 * f(0.25), -(g0 + g1 + g2) / f(6), -(g0 - g1 + g2) / f(6), g0 / f(24) + g1 / f(12) + g2 / f(6),
         g0 / f(24) - g1 / f(12) + g2 / f(6), g2]
    m = [u[k] @ v[k] for k in range(6)]
    y = np.empty((w.shape[0], 4 * Lq), f)
    y[:, 0::4] = m[0] + (m[1] + m[2]) + (m[3] + m[4]); y[:, 1::4] = (m[1] - m[2]) + f(2) * (m[3] - m[4])
    y[:, 2::4] = (m[1] + m[2]) + f(4) * (m[3] + m[4]); y[:, 3::4] = (m[1] - m[2]) + f(8) * (m[3] - m[4]) + m[5]
    return y[:, :L]


rng = np.random.default_rng(0)
print("C      form     rel L2     max / output scale")
for C, L in ((64, 500), (128, 250), (256, 125), (512, 63)):
    x = np.maximum(rng.standard_normal((C, L)), 0).astype(np.float32)
    w = (rng.standard_normal((C, C, 3)) * np.sqrt(2.0 / (3 * C))).astype(np.float32)
    ref = direct(x, w, np.float64)
    sc = np.abs(ref).max()
    for name, y in (("direct", direct(x, w, np.float32)), ("F(2,3)", wino23(x, w)), ("F(4,3)", wino43(x, w))):
        e = y.astype(np.float64) - ref
        print(f"{C:<6d} {name:8s} {np.linalg.norm(e) / np.linalg.norm(ref):.2e}   {np.abs(e).max() / sc:.2e}")
