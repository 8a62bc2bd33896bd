#!/usr/bin/env python3
"""Timing of the bf16 weight gradient (csrc/amp.hip: conv_wgrad_b16s1_kernel + wgrad_b16_reduce_kernel) at the bench's student batch;
back-to-back protocol of tools/ws_bench.py.  usage: python tools/wgrad_bench.py [N]   (WS_LAYERS=CinxLxCout[xKxstride],...)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semi-seg-ecg_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

from ssecg import amp as SAMP  # noqa: E402
from ssecg import ops  # noqa: E402
from ws_bench import LAYERS, timeit  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
for lay in LAYERS:
    Cin, L, Cout = lay[:3]
    K, st = (lay[3], lay[4]) if len(lay) > 3 else (3, 1)
    pad = K // 2
    Lo = ops.conv_out_len(L, K, st, pad, 1)
    x = SAMP.to_blocked(torch.randn(N, Cin, L, device=dev))
    dy = SAMP.to_blocked(torch.randn(N, Cout, Lo, device=dev))
    fl = 2.0 * N * Lo * Cout * Cin * K
    byt = 2.0 * (x.numel() + dy.numel())
    t, t0 = timeit(lambda: SAMP.conv_wgrad(dy, x, K, st, pad))
    print(f"{Cin:4d} {L:4d} {Cout:4d} k{K} s{st} | wgrad + reduce {t:6.1f} us (min {t0:6.1f}; {fl / t / 1e6:5.0f} TF, {byt / t / 1e3:5.0f} GB/s)", flush=True)
