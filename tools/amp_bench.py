#!/usr/bin/env python3
"""Per-layer timing of the bf16 kernels (csrc/amp.hip) at the bench's student batch (N = 1024): forward, data gradient,
weight gradient, BN passes.  HIP events on the launch stream, median of several rounds, random data.
usage: python tools/amp_bench.py [N]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semi-seg-ecg_amd"))
import torch  # noqa: E402

from ssecg import amp as SAMP  # noqa: E402
from ssecg import ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
LAYERS = [(64, 500, 64, 3, 1), (64, 500, 128, 3, 2), (64, 500, 128, 1, 2), (128, 250, 128, 3, 1), (128, 250, 256, 3, 2),
          (256, 125, 256, 3, 1), (256, 125, 512, 3, 2), (512, 63, 512, 3, 1), (512, 63, 128, 3, 1)]


def timeit(fn, rounds=7, reps=10):
    """median us per launch of ``reps`` back-to-back launches between ONE HIP event pair, ~1 ms of queued work in front so the
    host is never the limit (round 4: an event pair around every single launch - the round-2/3 protocol - adds ~10 us of marker
    latency to a 30-60 us kernel; rocprofv3 --kernel-trace durations agree with this protocol within 1 us, tools/ws_prof.sh)"""
    fn(); torch.cuda.synchronize()
    filler = torch.empty(1 << 26, device=dev)
    ts = []
    for _ in range(rounds):
        for _ in range(12):
            filler.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    ts.sort()
    return ts[len(ts) // 2]


def roof_us(flops, nbytes):
    return max(flops / 2.5e15, nbytes / 8.0e12) * 1e6


print(f"N={N}: Cin L Cout K s | fwd us (TF, GB/s, fraction of its own roof) | dgrad us (TF, frac) | wgrad us (TF, frac) | bn_apply / bwd_reduce / bwd_apply GB/s")
print("# roof of a launch = max(FLOPs / 2.5 PF, activation bytes in + out / 8 TB/s); wgrad bytes = both operands read once")
for Cin, L, Cout, K, s in LAYERS:
    pad = 1 if K == 3 else 0
    x = SAMP.to_blocked(torch.randn(N, Cin, L, device=dev))
    w = torch.randn(Cout, Cin, K, device=dev) * (2.0 / (K * Cout)) ** 0.5
    Lo = ops.conv_out_len(L, K, s, pad, 1)
    dy = SAMP.to_blocked(torch.randn(N, Cout, Lo, device=dev))
    ops.begin_forward()
    SAMP.conv_fwd(x, w, s, pad); SAMP.conv_dgrad(dy, w, L, s, pad)
    fl = 2.0 * N * Lo * Cout * Cin * K
    t_f = timeit(lambda: SAMP.conv_fwd(x, w, s, pad))
    t_d = timeit(lambda: SAMP.conv_dgrad(dy, w, L, s, pad))
    t_w = timeit(lambda: SAMP.conv_wgrad(dy, x, K, s, pad))
    byt = 2.0 * (x.numel() + dy.numel())
    mean = torch.zeros(Cout, device=dev); inv = torch.ones(Cout, device=dev); g = torch.ones(Cout, device=dev); b = torch.zeros(Cout, device=dev)
    t_a = timeit(lambda: SAMP.bn_apply_fwd(dy, mean, inv, g, b, None, True))
    t_r = timeit(lambda: SAMP.bn_bwd_reduce(dy, None, dy, mean, inv, g, b, 2))
    sums = torch.zeros(Cout, 2, device=dev, dtype=torch.float64)
    t_b = timeit(lambda: SAMP.bn_bwd_apply(dy, None, dy, mean, inv, g, b, 2, sums, N * Lo))
    nb = 2.0 * dy.numel()
    rf = roof_us(fl, byt)
    print(f"{Cin:4d} {L:4d} {Cout:4d} {K} {s} | {t_f:7.1f} ({fl / t_f / 1e6:6.0f} TF, {byt / t_f / 1e3:5.0f} GB/s, {rf / t_f:4.2f}) | {t_d:7.1f} ({fl / t_d / 1e6:6.0f}, {rf / t_d:4.2f}) | "
          f"{t_w:7.1f} ({fl / t_w / 1e6:6.0f}, {rf / t_w:4.2f}) | {2 * nb / t_a / 1e3:5.0f} / {2 * nb / t_r / 1e3:5.0f} / {3 * nb / t_b / 1e3:5.0f}", flush=True)
