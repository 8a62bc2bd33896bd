#!/usr/bin/env python3
"""Which host-side ATen calls of one FixMatch step launch the small fill / copy kernels?  torch.profiler with Python stacks:
prints, for aten::copy_ / fill_ / zero_ / zeros / clone, the innermost repo frame that issued each call and how often."""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semi-seg-ecg_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from torch.profiler import ProfilerActivity, profile

from helpers import TRAIN_CFG, build_hip_model, to_dev
from ssecg import synth
import algorithms.fixmatch as A_fm
from utils.misc import NativeScalerWithGradNormCount
from utils.optimizer import get_optimizer_from_config

B, C, L = int(sys.argv[1]) if len(sys.argv) > 1 else 16, 1, 2000
dev = torch.device("cuda:0")
model = build_hip_model(C, synth.model_state(0, C), dev)
opt = get_optimizer_from_config(dict(TRAIN_CFG), model.parameters())
scaler = NativeScalerWithGradNormCount()
batch = to_dev(synth.fixmatch_batch(1, B, C, L), dev)


def step():
    loss, stats = A_fm.fixmatch_step(model, batch["labeled"]["ecg"], batch["labeled"]["target"], batch["unlabeled"]["ecg"],
                                     batch["unlabeled"]["ecg_aug"], 0.8)
    scaler(loss, opt, clip_grad=None, parameters=model.parameters(), update_grad=True)
    opt.zero_grad()


for _ in range(3):
    step()
torch.cuda.synchronize()
import traceback
from torch.utils._python_dispatch import TorchDispatchMode

cnt = collections.Counter()
WATCH = ("copy_", "_to_copy", "fill_", "zero_", "zeros", "clone", "zeros_like", "empty_strided", "ones", "full", "cat", "add_", "mul",
         "_foreach_add_", "sum", "stack")


class Census(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name in WATCH:
            frames = [f for f in traceback.extract_stack() if "/repo/" in f.filename and "op_census" not in f.filename]
            where = f"{frames[-1].filename.replace(ROOT, '')}:{frames[-1].lineno} {frames[-1].name}" if frames else "<engine>"
            shapes = [tuple(a.shape) for a in args if isinstance(a, torch.Tensor)][:2]
            cnt[(name, where, str(shapes)[:60])] += 1
        return func(*args, **(kwargs or {}))


with Census():
    step()
    torch.cuda.synchronize()
for (name, where, shapes), n in cnt.most_common(60):
    print(f"{n:4d}  {name:14s} {where:70s} {shapes}")
