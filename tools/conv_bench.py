#!/usr/bin/env python3
"""Micro-benchmark of the conv kernels on the hot path's layer shapes (HIP events on the launch stream).
usage: python tools/conv_bench.py [fwd|dgrad|wgrad|all] [N] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semi-seg-ecg_amd")):
    sys.path.insert(0, p)
import torch
from ssecg import ops

SHAPES = [  # name, Cin, Lin, Cout, K, stride, pad
    ("stem   12->64  k7s2 L2000", 12, 2000, 64, 7, 2, 3),
    ("l1     64->64  k3   L500 ", 64, 500, 64, 3, 1, 1),
    ("l2.0   64->128 k3s2 L500 ", 64, 500, 128, 3, 2, 1),
    ("l2    128->128 k3   L250 ", 128, 250, 128, 3, 1, 1),
    ("l3.0  128->256 k3s2 L250 ", 128, 250, 256, 3, 2, 1),
    ("l3    256->256 k3   L125 ", 256, 125, 256, 3, 1, 1),
    ("l4.0  256->512 k3s2 L125 ", 256, 125, 512, 3, 2, 1),
    ("l4    512->512 k3   L63  ", 512, 63, 512, 3, 1, 1),
    ("l4ds  256->512 k1s2 L125 ", 256, 125, 512, 1, 2, 0),
    ("head  512->128 k3   L63  ", 512, 63, 128, 3, 1, 1),
]


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    only = sys.argv[4] if len(sys.argv) > 4 else None
    if os.environ.get("SSECG_WINOGRAD") is not None:
        print("SSECG_WINOGRAD =", os.environ["SSECG_WINOGRAD"], "(TF = algorithmic direct-conv FLOPs / time)")
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    for name, Cin, Lin, Cout, K, s, p in SHAPES:
        if only and only not in name:
            continue
        x = torch.randn(N, Cin, Lin, device=dev)
        w = torch.randn(Cout, Cin, K, device=dev) * 0.05
        Lout = ops.conv_out_len(Lin, K, s, p)
        dy = torch.randn(N, Cout, Lout, device=dev)
        flops = 2.0 * N * Lout * Cout * Cin * K
        res = []
        ops.begin_forward()   # operands transformed once, then trusted (as inside a model forward)
        for mode in ("fwd", "dgrad", "wgrad"):
            if which not in ("all", mode):
                continue
            def run():
                if mode == "fwd":
                    ops.conv1d_fwd(x, w, s, p, 1, want_stats=True, w_cached=True)
                elif mode == "dgrad":
                    ops.conv1d_dgrad(dy, w, Lin, s, p, 1, w_cached=True)
                else:
                    ops.conv1d_wgrad(dy, x, K, s, p, 1)
            run(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            res.append(f"{mode} {ms:7.3f} ms {flops / ms / 1e9:6.1f} TF")
            from ssecg.lib import lib
            if hasattr(lib(), "ssecg_debug_w4_clock"):   # diagnostic build (tools/ablate_wino4.sh CLOCK): in-kernel shader clock
                import ctypes
                f = lib().ssecg_debug_w4_clock; f.restype = ctypes.c_double; f.argtypes = [ctypes.c_int]
                res[-1] += f" clk {f(256):.3f} GHz"
        print(f"{name} N={N}  " + " | ".join(res), flush=True)


if __name__ == "__main__":
    main()
