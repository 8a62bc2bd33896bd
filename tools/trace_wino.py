#!/usr/bin/env python3
"""Debug: per-stage clock stamps of one wave of the Winograd kernel (needs a -DSSECG_WINO_TRACE build, see
tools/trace_wino.sh).  Prints, for stages of the first tile: cycles spent issuing loads, in the MFMA section,
in wait+LDS stores, at the barrier."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semi-seg-ecg_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
from ssecg import ops
from ssecg.lib import lib

N, C, L, M = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (1024, 512, 63, 512)
dev = torch.device("cuda:0")
x = torch.randn(N, C, L, device=dev); w = torch.randn(M, C, 3, device=dev) * 0.05
for _ in range(3):
    ops.conv1d_fwd(x, w, 1, 1, 1, want_stats=True)
torch.cuda.synchronize()
n = (C // 8) * 5
buf = (ctypes.c_ulonglong * n)()
fn = lib().ssecg_debug_wino_trace
fn.restype = ctypes.c_int; fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert fn(buf, n) == 0
t = np.array(buf[:], dtype=np.int64).reshape(-1, 5)
d = np.diff(t, axis=1)
print("stage: loads-issue  mfma-section  wait+store  barrier | total   (memtime ticks, 100 MHz => x24 core cycles at 2.4 GHz)")
for s in range(min(len(t) - 1, 40)):
    print(f"{s:3d}: {d[s,0]:6d} {d[s,1]:6d} {d[s,2]:6d} {d[s,3]:6d} | {t[s+1,0]-t[s,0]:6d}")
print("mean over stages:", d[1:-1].mean(axis=0), "stage period", np.diff(t[:, 0])[1:-1].mean())
