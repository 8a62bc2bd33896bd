"""Probe: forward + backward of one FixMatch step (B=16, C=12) captured in a HIP graph (torch.cuda.graph over the ctypes
launches) vs eager.  Round-1 result on one MI355X: replay 5.97 ms, eager 6.60 ms - at launch-bound-looking batch sizes the
step is bound by the small kernels' own duration (persistent grids sized for the chip), not by dispatch: graph capture
buys 10 %, so it was not built into the training loop."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semi-seg-ecg_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from helpers import TRAIN_CFG, build_hip_model, to_dev
from ssecg import synth, functional as SF
import algorithms.fixmatch as A_fm

dev = torch.device("cuda:0")
B, C, L = 16, 12, 2000
model = build_hip_model(C, synth.model_state(0, C), dev)
model.decode_head.dropout = None; model.decode_head.dropout_ratio = 0.0
batch = to_dev(synth.fixmatch_batch(1, B, C, L), dev)
args = (batch["labeled"]["ecg"], batch["labeled"]["target"], batch["unlabeled"]["ecg"], batch["unlabeled"]["ecg_aug"])

def step():
    loss, stats = A_fm.fixmatch_step(model, *args, 0.8)
    loss.backward()
    return loss, stats

s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        for p in model.parameters(): p.grad = None
        step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
for p in model.parameters(): p.grad = None
with torch.cuda.graph(g):
    loss, stats = step()
torch.cuda.synchronize()
print("captured; loss", float(loss), stats.tolist())
g.replay(); torch.cuda.synchronize()
print("replayed; loss", float(loss), stats.tolist())
t0 = time.perf_counter()
for _ in range(50): g.replay()
torch.cuda.synchronize()
print(f"graph replay fwd+bwd: {(time.perf_counter() - t0) / 50 * 1e3:.2f} ms/step")
t0 = time.perf_counter()
for _ in range(50):
    for p in model.parameters(): p.grad = None
    step()
torch.cuda.synchronize()
print(f"eager fwd+bwd:        {(time.perf_counter() - t0) / 50 * 1e3:.2f} ms/step")
