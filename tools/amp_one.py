#!/usr/bin/env python3
"""One bf16 conv shape, forward + data gradient, for PMC runs.  usage: python tools/amp_one.py [Cin L Cout N reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semi-seg-ecg_amd"))
import torch
from ssecg import amp as SAMP, ops
a = [int(v) for v in sys.argv[1:]]
Cin, L, Cout, N, reps = (a + [512, 63, 512, 1024, 5][len(a):])[:5]
dev = torch.device("cuda:0")
x = SAMP.to_blocked(torch.randn(N, Cin, L, device=dev))
w = torch.randn(Cout, Cin, 3, device=dev) * (2.0 / (3 * Cout)) ** 0.5
dy = SAMP.to_blocked(torch.randn(N, Cout, L, device=dev))
ops.begin_forward()
for _ in range(reps):
    SAMP.conv_fwd(x, w, 1, 1, want_stats=True)
    SAMP.conv_dgrad(dy, w, L, 1, 1)
torch.cuda.synchronize()
