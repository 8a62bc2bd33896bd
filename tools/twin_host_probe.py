#!/usr/bin/env python3
"""How far is oracle/torch_ref.py on THIS host's CPU from the reference outputs frozen in the stepfix_* fixtures (bit-identical
in the build container)?  Worst gradient statistic per fixture and step, for the default thread count and for 8 threads."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "semi-seg-ecg_amd"), os.path.join(ROOT, "tests")]
import torch

from helpers import StepfixTwin, check_rows, golden

for nt in (torch.get_num_threads(), 8, 1):
    torch.set_num_threads(nt)
    for name in ("stepfix_fixmatch_c12_b2_L250", "stepfix_cps_c2_b1_L250", "stepfix_base_c1_b4_L250"):
        g = golden(name)
        tw = StepfixTwin(g)
        for s in range(tw.nsteps):
            r = tw.step(s)
            w = check_rows(g, f"step{s}.grad.", r["grads"], 1.0, what="twin")
            dl = (r["logits"] - torch.from_numpy(g[f"step{s}.logits"])).abs().max().item()
            print(f"threads {nt:3d} {name} step {s}: worst gradient statistic {w:.2e}, logits max|d| {dl:.2e}", flush=True)
