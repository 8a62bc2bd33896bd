#!/usr/bin/env python3
"""Reads the s_memtime sums a -DSSECG_WS_STAMP build of csrc/amp_ws.hip leaves in the statistics rows (diagnostic build only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "semi-seg-ecg_amd"))
import numpy as np, torch
from ssecg import amp as SAMP, ops
dev = torch.device("cuda:0"); N = 1024
for Cin, L, Cout in [(64, 500, 64), (128, 250, 128), (256, 125, 256)]:
    x = SAMP.to_blocked(torch.randn(N, Cin, L, device=dev)); w = torch.randn(Cout, Cin, 3, device=dev) * 0.05
    ops.begin_forward()
    for _ in range(3):
        y, st = SAMP.conv_fwd(x, w, 1, 1)
    torch.cuda.synchronize()
    raw = st.cpu().numpy().view(np.uint64).reshape(st.shape[0], -1)      # per row: Cout u64; group g at offset 32 g
    MG = max(1, Cout // 128)
    rows = np.stack([raw[:, 32 * g: 32 * g + 12] for g in range(MG)]).reshape(-1, 12).astype(np.float64)
    med = np.median(rows, axis=0)
    names = ["loop total", "vmcnt wait", "barrier", "stage body", "tile setup", "tile end", "W load", "prologue", "flush", "drain", "kernel", "realtime(100MHz)"]
    print(f"{Cin}x{L}->{Cout}: " + "  ".join(f"{n} {v:9.0f}" for n, v in zip(names, med)) + f"   clock " + f"{med[10] / max(med[11], 1) * 0.1:.2f}" + f" GHz (cycles per workgroup; {rows.shape[0]} workgroups)")
