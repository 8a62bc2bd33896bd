#!/usr/bin/env python3
"""Stem kernels (csrc/stem.hip) vs the generic implicit GEMM on the bench shape: forward (+BN statistics), eval-mode fused
conv+BN+ReLU+pool vs the two-launch chain, weight gradient.  HIP events, same process.  usage: python tools/stem_bench.py [N]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semi-seg-ecg_amd"))
import torch  # noqa: E402

from ssecg import ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
C, L = 12, 2000
dev = torch.device("cuda:0")
x = torch.randn(N, C, L, device=dev)
w = torch.randn(64, C, 7, device=dev) * 0.1
dc = torch.randn(N, 64, 1000, device=dev)
scale, shift = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.1


def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


flops = 2.0 * N * 1000 * 64 * C * 7
for stem in (True, False):
    ops.STEM = stem
    name = "stem.hip" if stem else "generic "
    f = t(lambda: ops.conv1d_fwd(x, w, 2, 3, 1, want_stats=True))
    g = t(lambda: ops.conv1d_wgrad(dc, x, 7, 2, 3, 1))
    if stem:
        e = t(lambda: ops.stem_fwd_eval_pool(x, w, scale, shift))
    else:
        e = t(lambda: ops.bn_relu_maxpool_fwd(ops.conv1d_fwd(x, w, 2, 3, 1)[0], None, None, scale, shift, 3, 2, 1))
    print(f"{name} N={N}: fwd+stats {f:7.1f} us ({flops / f / 1e6:5.1f} TF, {(x.numel() + N * 64000) * 4 / f / 1e3:5.0f} GB/s) | "
          f"eval conv+BN+ReLU+pool {e:7.1f} us | wgrad {g:7.1f} us ({flops / g / 1e6:5.1f} TF)", flush=True)
