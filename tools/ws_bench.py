#!/usr/bin/env python3
"""Forward / data-gradient timing of the 3-tap stride-1 bf16 convolutions only (csrc/amp_ws.hip, csrc/amp.hip ring kernel) at the
bench's student batch; HIP events around each launch, median of several rounds, random data.  usage: python tools/ws_bench.py [N]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semi-seg-ecg_amd"))
import torch  # noqa: E402

from ssecg import amp as SAMP  # noqa: E402
from ssecg import ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
LAYERS = [(64, 500, 64), (128, 250, 128), (256, 125, 256), (512, 63, 512), (512, 63, 128),
          (64, 500, 128, 3, 2), (64, 500, 128, 1, 2), (128, 250, 256, 3, 2), (128, 250, 256, 1, 2), (256, 125, 512, 3, 2), (256, 125, 512, 1, 2)]
if os.environ.get("WS_LAYERS"):
    LAYERS = [tuple(int(v) for v in t.split("x")) for t in os.environ["WS_LAYERS"].split(",")]   # CinxLxCout[xKxstride]


def timeit(fn, rounds=7, reps=10):
    """us per launch of `reps` back-to-back launches between ONE event pair (an event pair per launch adds ~10 us of marker
    latency to a 30 us kernel; the kernel boundary of a back-to-back stream is part of what a training step pays)."""
    fn(); torch.cuda.synchronize()
    filler = torch.empty(1 << 26, device=dev)
    ts = []
    for _ in range(rounds):
        for _ in range(12):
            filler.zero_()      # ~1 ms of queued work: the host enqueues the timed launches while the GPU is still busy with this
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


if __name__ == "__main__":
    for lay in LAYERS:
        Cin, L, Cout = lay[:3]
        K, st = (lay[3], lay[4]) if len(lay) > 3 else (3, 1)
        pad = K // 2
        Lo = ops.conv_out_len(L, K, st, pad, 1)
        x = SAMP.to_blocked(torch.randn(N, Cin, L, device=dev))
        w = torch.randn(Cout, Cin, K, device=dev) * (2.0 / (K * Cout)) ** 0.5
        dy = SAMP.to_blocked(torch.randn(N, Cout, Lo, device=dev))
        ops.begin_forward()
        SAMP.conv_fwd(x, w, st, pad); SAMP.conv_dgrad(dy, w, L, st, pad)
        fl = 2.0 * N * Lo * Cout * Cin * K
        byt = 2.0 * (x.numel() + dy.numel())
        tf, tf0 = timeit(lambda: SAMP.conv_fwd(x, w, st, pad))
        td, td0 = timeit(lambda: SAMP.conv_dgrad(dy, w, L, st, pad))
        print(f"{Cin:4d} {L:4d} {Cout:4d} k{K} s{st} | fwd {tf:6.1f} us (min {tf0:6.1f}; {fl / tf / 1e6:5.0f} TF, {byt / tf / 1e3:5.0f} GB/s) | dgrad {td:6.1f} us (min {td0:6.1f}; {fl / td / 1e6:5.0f} TF)", flush=True)
