import sys, os, time
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "semi-seg-ecg_amd")); sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from algorithms.base import init_model_from_cfg
from ssecg import functional as SF
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
from helpers import model_cfg
dev = torch.device("cuda:0")
m = init_model_from_cfg(model_cfg(12)).to(dev).eval()
x = torch.randn(256, 12, 2000, device=dev); y = torch.randint(0, 4, (256, 2000), device=dev)
def t(f, name):
    torch.cuda.synchronize(); t0 = time.time(); r = f(); torch.cuda.synchronize(); print(f"{name}: {(time.time()-t0)*1e3:.1f} ms"); return r
with torch.no_grad():
    for it in range(2):
        res = t(lambda: m(x, y, return_loss=True), "forward")
        logits = res["seg_logits"]
        _, pred, prob = t(lambda: SF.pseudo_label(logits, want_prob=True), "pseudo_label")
        cm = t(lambda: torch.bincount((y * 4 + pred).reshape(-1), minlength=16).reshape(4, 4), "bincount")
        t(lambda: res["loss"].item(), "loss.item")
        t(lambda: prob.cpu(), "prob.cpu")
