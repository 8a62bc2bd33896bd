import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semi-seg-ecg_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
from helpers import *
from ssecg import synth
dev = torch.device("cuda:0")
C, B, seed = int(sys.argv[1]), 2, int(sys.argv[2])
g = golden(f"forward_c{C}_b{B}")
model = build_hip_model(C, synth.model_state(seed, C, trained=True, sharpen=sharpen_for(C)), dev)
x = torch.from_numpy(synth.normal(seed + 1, 1, (B, C, L))).to(dev)
y = torch.from_numpy(synth.labels(seed + 1, 4, B, L)).to(dev)
model.train()
model.decode_head.fixed_dropout_mask = torch.from_numpy(dropout_mask_np(seed + 1, B)).to(dev, torch.uint8)
res = model(x, y, return_loss=True)
res["loss"].backward()
from ssecg.functional import wait_for_wgrads
wait_for_wgrads()
rows = []
for k, p in model.named_parameters():
    fk = "train.grad.full." + k
    if fk in g.files:
        ref = torch.from_numpy(g[fk]); got = p.grad.cpu()
        d = (got - ref).abs()
        rows.append((d.max().item() / ref.abs().max().item(), k, int(d.argmax()), int((d > 0.01 * ref.abs().max()).sum()), ref.numel()))
rows.sort(reverse=True)
for r in rows[:25]:
    print("%.3e %-40s argmax=%d n_bad=%d / %d" % r)
k = rows[0][1]
ref = torch.from_numpy(g["train.grad.full." + k]); got = dict(model.named_parameters())[k].grad.cpu()
idx = (got - ref).abs().flatten().topk(5).indices
print(k, [(int(i), float(got.flatten()[i]), float(ref.flatten()[i])) for i in idx])
