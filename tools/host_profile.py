#!/usr/bin/env python3
"""Host-side cost of one FixMatch step at a launch-bound batch size (cProfile, top functions by cumulative time)."""
import cProfile, os, pstats, sys, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semi-seg-ecg_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from helpers import TRAIN_CFG, build_hip_model, to_dev
from ssecg import synth
import algorithms.fixmatch as A_fm
from utils.optimizer import get_optimizer_from_config

B, C, L = int(sys.argv[1]) if len(sys.argv) > 1 else 16, 12, 2000
dev = torch.device("cuda:0")
model = build_hip_model(C, synth.model_state(0, C), dev)
opt = get_optimizer_from_config(dict(TRAIN_CFG), model.parameters())
batch = to_dev(synth.fixmatch_batch(1, B, C, L), dev)


def step():
    loss, stats = A_fm.fixmatch_step(model, batch["labeled"]["ecg"], batch["labeled"]["target"], batch["unlabeled"]["ecg"],
                                     batch["unlabeled"]["ecg_aug"], 0.8)
    loss.backward()
    opt.step(); opt.zero_grad()


for _ in range(5):
    step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(20):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"B={B}: host enqueue {1e3 * (t1 - t0) / 20:.2f} ms/step, + drain {1e3 * (t2 - t1):.2f} ms total")
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    step()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18); print(s.getvalue()[:3500])
