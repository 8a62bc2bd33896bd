#!/usr/bin/env python3
"""fp32 Winograd weight gradient, transpose of F(4,3) against transpose of F(2,3), at the bench's student batch; back-to-back
protocol of tools/ws_bench.py.  usage: python tools/wino_wgrad_bench.py [N]   (WS_LAYERS=CinxLxCout,...)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semi-seg-ecg_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

from ssecg import ops  # noqa: E402
from ws_bench import timeit  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
LAYERS = [(64, 500, 64), (128, 250, 128), (256, 125, 256), (512, 63, 512), (512, 63, 128)]
if os.environ.get("WS_LAYERS"):
    LAYERS = [tuple(int(v) for v in t.split("x")) for t in os.environ["WS_LAYERS"].split(",")]
dev = torch.device("cuda:0")
for Cin, L, Cout in LAYERS:
    x = torch.randn(N, Cin, L, device=dev).relu_()
    dy = torch.randn(N, Cout, L, device=dev) * 0.01
    aff = (torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.1)
    fl = 2.0 * N * L * Cout * Cin * 3
    ref = None
    for wf in (2, 4):
        ops.WINO_WGRAD_F = wf
        t, t0 = timeit(lambda: ops.conv1d_wgrad(dy, x, 3, 1, 1, 1))
        ta, _ = timeit(lambda: ops.conv1d_wgrad(dy, x, 3, 1, 1, 1, x_affine=aff))
        dw = ops.conv1d_wgrad(dy, x, 3, 1, 1, 1).double()
        if ref is None:
            ref = torch.nn.grad.conv1d_weight(x.double(), (Cout, Cin, 3), dy.double(), padding=1) if N * L * Cin * Cout < 2e11 else dw
        err = ((dw - ref).abs().max() / ref.abs().max()).item()
        print(f"{Cin:4d} {L:4d} {Cout:4d} | transpose of F({wf},3): kernel + reduce {t:6.1f} us (min {t0:6.1f}; {fl / t / 1e6:5.0f} TF direct-equivalent), "
              f"with fused input BN + ReLU {ta:6.1f} us, max err / scale {err:.2e}", flush=True)
