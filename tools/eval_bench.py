#!/usr/bin/env python3
"""Throughput of algorithms.base.evaluate() on synthetic validation batches resident on the host, as a
loader hands them over: eval-mode forward (BN folded), argmax, per-record confusion counts.  Times the fast path every train()
loop uses (return_outputs=False) and the full path test() uses (probabilities + one-hot labels copied to the host).
usage: python tools/eval_bench.py [batch] [n_batches] [leads] [length]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "semi-seg-ecg_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
from algorithms.base import evaluate, init_model_from_cfg  # noqa: E402
from ssecg import synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 8
C = int(sys.argv[3]) if len(sys.argv) > 3 else 12
L = int(sys.argv[4]) if len(sys.argv) > 4 else 2000
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = init_model_from_cfg(bench.model_config(C)).to(dev)
one = synth.fixmatch_batch(77, B, C, L)["labeled"]
loader = [{"ecg": torch.from_numpy(one["ecg"]).pin_memory(), "target": torch.from_numpy(one["target"]).pin_memory()} for _ in range(NB)]
dev_loader = [{k: v.to(dev) for k, v in b.items()} for b in loader]
import contextlib  # noqa: E402
import io  # noqa: E402

for amp in (False, True):     # use_amp: the reference runs evaluate() inside autocast (base.py:202): the 16-bit eval path (round 6)
    for name, ld, ro in (("fast path, inputs in HBM", dev_loader, False), ("fast path, host batches (PCIe inclusive)", loader, False),
                         ("full outputs to the host (test())", loader, True)):
        with contextlib.redirect_stdout(io.StringIO()):
            evaluate(model, ld[:2], dev, None, use_amp=amp, return_outputs=ro)
            torch.cuda.synchronize()
            dt = float("inf")
            for _ in range(3):        # (the first pass of a precision also pays the allocator's first blocks of its shapes)
                t0 = time.perf_counter()
                evaluate(model, ld, dev, None, use_amp=amp, return_outputs=ro)
                torch.cuda.synchronize()
                dt = min(dt, time.perf_counter() - t0)
        print(f"evaluate(use_amp={amp}) {name}: B={B} x {NB} batches, C={C}, L={L}: {dt / NB * 1e3:.2f} ms/batch = {B * NB / dt:,.0f} windows/s",
              flush=True)
