#!/usr/bin/env python3
"""Step time of every plugin on synthetic batches (HBM-resident inputs, one MI355X): base (supervised), fixmatch, mean_teacher
(EMA update included), cps (two models), stpp (frozen teacher) - the same fused units, different step bodies.
usage: python tools/plugin_bench.py [B] [C] [steps] [--amp]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "semi-seg-ecg_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402
import utils.lr_sched as lr_sched  # noqa: E402
from algorithms.base import init_model_from_cfg  # noqa: E402
from algorithms.cps import cps_loss, cps_pseudo_labels  # noqa: E402
from algorithms.fixmatch import fixmatch_step  # noqa: E402
from algorithms.mean_teacher import mean_teacher_step  # noqa: E402
from algorithms.stpp import stpp_step  # noqa: E402
from ssecg.optim import EmaUpdater  # noqa: E402
from utils.misc import NativeScalerWithGradNormCount  # noqa: E402
from utils.optimizer import get_optimizer_from_config  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    amp = "--amp" in sys.argv
    B = int(args[0]) if len(args) > 0 else 256
    C = int(args[1]) if len(args) > 1 else 12
    steps = int(args[2]) if len(args) > 2 else 20
    dev = torch.device("cuda:0")
    cfg = dict(bench.TRAIN_CFG)
    ecg_x, mask_x, ecg_u_w, ecg_u_s = bench.synthetic_batch(B, C, 2000, 1234, dev)

    def model():
        torch.manual_seed(0)
        m = init_model_from_cfg(bench.model_config(C)).to(dev)
        if amp:
            from ssecg import amp as SAMP
            SAMP.enable(m)
        return m

    def run(name, make):
        step = make()
        for i in range(5):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(5 + i)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        print(f"{name:13s} B={B} C={C} {'bf16' if amp else 'fp32'}: {ms:7.3f} ms/step  {B / ms * 1e3:9.0f} windows/s", flush=True)

    def mk_base():
        m = model(); opt = get_optimizer_from_config(cfg, m.parameters()); sc = NativeScalerWithGradNormCount()

        def step(i):
            lr_sched.adjust_learning_rate(opt, 10.0 + i / 1000.0, cfg)
            loss = m(ecg_x, mask_x, return_loss=True)['loss']
            sc(loss, opt, clip_grad=None, parameters=m.parameters(), update_grad=True); opt.zero_grad()
        m.train()
        return step

    def mk_fixmatch():
        m = model(); opt = get_optimizer_from_config(cfg, m.parameters()); sc = NativeScalerWithGradNormCount()

        def step(i):
            lr_sched.adjust_learning_rate(opt, 10.0 + i / 1000.0, cfg)
            loss, _ = fixmatch_step(m, ecg_x, mask_x, ecg_u_w, ecg_u_s, cfg['conf_thresh'])
            sc(loss, opt, clip_grad=None, parameters=m.parameters(), update_grad=True); opt.zero_grad()
        return step

    def mk_mt():
        m = model(); t = model()
        for p in t.parameters():
            p.requires_grad = False
        t.eval()
        opt = get_optimizer_from_config(cfg, m.parameters()); sc = NativeScalerWithGradNormCount(); ema = EmaUpdater()

        def step(i):
            lr_sched.adjust_learning_rate(opt, 10.0 + i / 1000.0, cfg)
            loss, _ = mean_teacher_step(m, t, ecg_x, mask_x, ecg_u_w, ecg_u_s)
            sc(loss, opt, clip_grad=None, parameters=m.parameters(), update_grad=True); opt.zero_grad()
            ema(m, t, 0.99)
        return step

    def mk_cps():
        m1, m2 = model(), model()
        o1, o2 = get_optimizer_from_config(cfg, m1.parameters()), get_optimizer_from_config(cfg, m2.parameters())
        sc = NativeScalerWithGradNormCount()
        m1.train(); m2.train()

        def step(i):
            lr_sched.adjust_learning_rate(o1, 10.0 + i / 1000.0, cfg); lr_sched.adjust_learning_rate(o2, 10.0 + i / 1000.0, cfg)
            k1, k2 = cps_pseudo_labels(m1, m2, ecg_u_w)
            for m, o, k in ((m1, o1, k2), (m2, o2, k1)):
                loss, _ = cps_loss(m, ecg_x, mask_x, ecg_u_w, k)
                sc(loss, o, clip_grad=None, parameters=m.parameters(), update_grad=True); o.zero_grad()
        return step

    def mk_stpp():
        m = model(); t = model()
        for p in t.parameters():
            p.requires_grad = False
        t.eval()
        opt = get_optimizer_from_config(cfg, m.parameters()); sc = NativeScalerWithGradNormCount()

        def step(i):
            lr_sched.adjust_learning_rate(opt, 10.0 + i / 1000.0, cfg)
            loss, _ = stpp_step(m, t, ecg_x, mask_x, ecg_u_w)
            sc(loss, opt, clip_grad=None, parameters=m.parameters(), update_grad=True); opt.zero_grad()
        return step

    for name, mk in (("base", mk_base), ("fixmatch", mk_fixmatch), ("mean_teacher", mk_mt), ("cps", mk_cps), ("stpp", mk_stpp)):
        run(name, mk)


if __name__ == "__main__":
    main()
