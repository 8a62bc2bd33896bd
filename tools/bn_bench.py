#!/usr/bin/env python3
"""BatchNorm backward reduction at the student pass's shapes (N = 1024 windows): the 4-byte path of rounds 1-5 (SSECG_BN_ROWS=0) against
the 16-byte raw-buffer-load rows kernel (round 6) for the row lengths that are not multiples of 4 (250, 125, 63), and the aligned
float4 path at L = 500 for reference.  usage (GPU box): python tools/bn_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semi-seg-ecg_amd"))
from ssecg import ops  # noqa: E402

dev = torch.device("cuda:0")
N = int(os.environ.get("N", "1024"))
print(f"bn_bwd_reduce, N = {N}: microseconds per launch (20 back-to-back launches between one event pair), algorithmic GB/s = 2 tensors read")
for C, L in ((64, 500), (128, 250), (256, 125), (512, 63), (128, 63)):
    g = torch.Generator(device=dev).manual_seed(C + L)
    x = torch.randn((N, C, L), generator=g, device=dev)
    dy = torch.randn((N, C, L), generator=g, device=dev)
    mean, invstd = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    y, mask = ops.bn_apply_fwd(x, mean, invstd, gam, bet, dy, True, want_mask=True)
    row = []
    for mode, kw in (("recompute", dict(y=None, gamma=gam, beta=bet, relu_recompute=True)), ("mask bits", dict(y=mask)), ("saved y", dict(y=y))):
        for rows in ("0", "1"):
            if rows == "0":
                os.environ["SSECG_BN_ROWS"] = "0"
            else:
                os.environ.pop("SSECG_BN_ROWS", None)
            for _ in range(3):
                ops.bn_bwd_reduce(dy, kw.get("y"), x, mean, invstd, kw.get("gamma"), kw.get("beta"), relu_recompute=kw.get("relu_recompute", False))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.bn_bwd_reduce(dy, kw.get("y"), x, mean, invstd, kw.get("gamma"), kw.get("beta"), relu_recompute=kw.get("relu_recompute", False))
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 20 * 1e3
            row.append(f"{mode} {'rows' if rows == '1' else 'old '} {us:6.1f} us {2 * x.numel() * 4 / us / 1e3:6.0f} GB/s")
    print(f"C={C:4d} L={L:4d}: " + " | ".join(row), flush=True)
os.environ.pop("SSECG_BN_ROWS", None)
