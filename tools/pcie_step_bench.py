#!/usr/bin/env python3
"""PCIe-inclusive rate of the FixMatch step: the loaders hand over pinned HOST batches and the plugin moves them with
``.to(device, non_blocking=True)`` on the compute stream (reference: src/algorithms/fixmatch.py:80-84), so a step = H2D of
3 x (B, C, L) fp32 windows + (B, L) int64 labels, then the step.  Reported beside the HBM-resident rate of bench.py; with
``--prefetch`` the next batch is copied on a side stream while the current step computes (what a DataLoader with
``pin_memory`` + a prefetching wrapper gives).  usage: python tools/pcie_step_bench.py [--batch 512] [--steps 20] [--amp]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semi-seg-ecg_amd")):
    sys.path.insert(0, p)
import torch

import bench as B_
import utils.lr_sched as lr_sched
from algorithms.base import init_model_from_cfg
from algorithms.fixmatch import fixmatch_step
from utils.misc import NativeScalerWithGradNormCount
from utils.optimizer import get_optimizer_from_config

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=512); ap.add_argument("--leads", type=int, default=12)
ap.add_argument("--steps", type=int, default=20); ap.add_argument("--warmup", type=int, default=5)
ap.add_argument("--amp", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda:0")
B, C, L = args.batch, args.leads, 2000
torch.manual_seed(0)
model = init_model_from_cfg(B_.model_config(C)).to(dev)
if args.amp:
    from ssecg import amp as SAMP
    SAMP.enable(model)
cfg = dict(B_.TRAIN_CFG)
opt = get_optimizer_from_config(cfg, model.parameters())
scaler = NativeScalerWithGradNormCount()
host = [tuple(t.cpu().pin_memory() for t in B_.synthetic_batch(B, C, L, 1234 + i, dev)) for i in range(3)]
nbytes = sum(t.numel() * t.element_size() for t in host[0])


def step(dev_batch, i):
    lr_sched.adjust_learning_rate(opt, 10.0 + i / 1000.0, cfg)
    loss, stats = fixmatch_step(model, *dev_batch, cfg["conf_thresh"])
    scaler(loss, opt, clip_grad=None, parameters=model.parameters(), update_grad=True)
    opt.zero_grad()


def run(mode, n, i0):
    side = torch.cuda.Stream()
    nxt = None
    for i in range(n):
        hb = host[(i0 + i) % 3]
        if mode == "resident":
            db = resident
        elif mode == "inline":
            db = tuple(t.to(dev, non_blocking=True) for t in hb)
        else:   # prefetch: batch i was requested on the side stream during step i-1
            if nxt is None:
                with torch.cuda.stream(side):
                    nxt = tuple(t.to(dev, non_blocking=True) for t in hb)
            torch.cuda.current_stream().wait_stream(side)
            db = nxt
            for t in db:
                t.record_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                nxt = tuple(t.to(dev, non_blocking=True) for t in host[(i0 + i + 1) % 3])
        step(db, i0 + i)


resident = tuple(t.to(dev) for t in host[0])
out = {"B": B, "C": C, "L": L, "dtype": "bf16" if args.amp else "f32", "host_bytes_per_step": nbytes}
for mode in ("resident", "inline", "prefetch"):
    run(mode, args.warmup, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(mode, args.steps, args.warmup)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    out[mode] = {"ms_per_step": dt * 1e3, "windows_per_s": B / dt}
print(json.dumps(out))
