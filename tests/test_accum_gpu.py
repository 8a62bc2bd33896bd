"""GPU: gradient accumulation and clipping THROUGH the plugin: ``accum_iter = 2`` and an active ``max_norm`` on
the FixMatch ``train_one_epoch``, against the reference's real loop run on the same loaders (tools/make_golden.py::gen_accum_case,
fixture ``accumfix_fixmatch_c12_b2_L250``: two optimiser steps = four micro-batches, each searched tie-free on the state it
sees - incl. the BN running statistics the previous micro-step's train-mode forward leaves for the next pseudo-label pass).

Reference semantics under test (src/algorithms/fixmatch.py:73-78,129-138; src/utils/misc.py:242-256,265-278): the lr is set on
the first iteration of an accumulation window; every loss is divided by accum_iter before its backward; gradients add up in
``.grad``; only the last iteration of the window unscales, clips to max_norm (``clip_grad_norm_``: scale by
max_norm / (norm + 1e-6) when the norm exceeds it), steps AdamW and zeroes the gradients.

Asserted per optimiser step: every micro-step's student / teacher logits <= 1e-4, arg-max and keep masks bit-exact; ALL 65
gradients of EACH micro-step's backward <= 1e-4 (row norms, row sums, projections, full small tensors; no flip tolerance);
the norm the scaler returns (the pre-clip norm of the accumulated gradient) <= 1e-4 of the reference's; BN buffers <= 1e-5;
the AdamW update element by element (helpers.check_update_elementwise) against the live oracle twin and, where the fixture
stores tensor + gradients in full, against the reference's stored update.  A loop that forgot the division, stepped on every
iteration, clipped each micro-gradient, or clipped after the step fails every one of the last three."""
import numpy as np
import pytest
import torch

from helpers import (TRAIN_CFG, adamw_cond, build_hip_model, check_packed, check_rows, check_update_elementwise, cpu_batch,
                     dropout_mask_np, golden, rel, to_dev)
from ssecg import functional as SF
from ssecg import synth

pytestmark = pytest.mark.gpu
TOL = 1e-4
NAME = "accumfix_fixmatch_c12_b2_L250"


class _Tap:
    """The product's scaler with the norm it returns recorded (the plugin, like the reference, discards it)."""

    def __init__(self):
        from utils.misc import NativeScalerWithGradNormCount
        self.inner, self.norms = NativeScalerWithGradNormCount(), []

    def __call__(self, *a, **k):
        n = self.inner(*a, **k)
        if n is not None:
            self.norms.append(n)
        return n

    def state_dict(self):
        return self.inner.state_dict()

    def load_state_dict(self, sd):
        return self.inner.load_state_dict(sd)


class AccumTwin:
    """Live oracle twin of the fixture (oracle/torch_ref.fixmatch_accum_step; bit-identical to the reference in the build
    container, re-checked against the stored statistics on this host by tests/test_oracle_golden.py)."""

    def __init__(self, g):
        from oracle import torch_ref as O
        self.O, self.g = O, g
        self.C, self.B, self.L, self.seed, self.feat_len, self.nsteps, self.accum = (int(v) for v in g["meta"])
        self.sd_np = synth.model_state(self.seed, self.C, trained=True, sharpen=1.0)
        self.o, self.opt = O.state_from_numpy(self.sd_np), {}
        self.cfg = dict(TRAIN_CFG, accum_iter=self.accum, max_norm=float(g["max_norm"]), conf_thresh=float(g["conf_thresh"]))
        self.ocfg = dict(self.cfg, betas=(0.9, 0.999))
        self.pnames = O.param_names(self.o)

    def epoch(self, s):
        return 3 + 9 * s

    def inputs(self, s):
        seeds = [int(v) for v in self.g[f"step{s}.bseeds"]]
        return ([synth.fixmatch_batch(b, self.B, self.C, self.L) for b in seeds],
                [dropout_mask_np(b, 2 * self.B, lp=self.feat_len) for b in seeds])

    def step(self, s):
        batches, dms = self.inputs(s)
        r = self.O.fixmatch_accum_step(self.o, self.opt, [cpu_batch(b) for b in batches], self.ocfg, self.epoch(s),
                                       [torch.from_numpy(d.astype(np.float32)) for d in dms])
        # sign-like elements of the post-step state <- the reference's values (helpers.StepfixTwin._patch, same reason)
        with torch.no_grad():
            for k in self.pnames:
                idx = torch.from_numpy(self.g[f"step{s}.fix.idx.{k}"].astype(np.int64))
                self.o[k].detach().reshape(-1)[idx] = torch.from_numpy(self.g[f"step{s}.fix.val.{k}"])
        return r


def test_accumulation_and_clipping_through_the_fixmatch_plugin(dev):
    import algorithms.fixmatch as A_fm
    from utils.optimizer import get_optimizer_from_config
    g = golden(NAME)
    tw = AccumTwin(g)
    assert tw.accum == 2 and tw.nsteps == 2 and float(g["max_norm"]) < 0.51 * float(g["unclipped_norm0"])   # the clip is active
    model = build_hip_model(tw.C, tw.sd_np, dev)
    cfg = dict(tw.cfg)
    opt = get_optimizer_from_config(cfg, model.parameters())
    tap = _Tap()
    calls, grads = [], {}
    model.register_forward_hook(lambda m, i, o: calls.append(o["seg_logits"].detach().clone()))
    for k, p in model.named_parameters():
        p.register_hook(lambda gr, k=k: grads.setdefault(k, []).append(gr.detach().clone()))
    for s in range(tw.nsteps):
        pre = f"step{s}."
        assert float(g[pre + "fp32_vs_fp64_rel_l2"]) <= 1e-5
        if s > 0:   # continue from the REFERENCE's state: the fixture's next micro-batches are tie-free for that state
            sd = model.state_dict()
            with torch.no_grad():
                for k, v in sd.items():
                    v.copy_(tw.o[k].detach().to(v.device))
            for k, p in model.named_parameters():
                st = opt.state[p]
                assert int(torch.as_tensor(st["step"]).item()) == tw.opt["step"]
                st["exp_avg"].copy_(tw.opt["exp_avg." + k].to(p.device))
                st["exp_avg_sq"].copy_(tw.opt["exp_avg_sq." + k].to(p.device))
        batches_np, dms = tw.inputs(s)
        batches = [to_dev(b, dev) for b in batches_np]
        model.decode_head.fixed_dropout_mask = [torch.from_numpy(d).to(dev, torch.uint8) for d in dms]   # one per micro-step
        calls.clear(); grads.clear(); tap.norms.clear()
        before = {k: p.detach().clone() for k, p in model.named_parameters()}
        r = tw.step(s)
        for m in range(tw.accum):
            check_rows(g, f"{pre}m{m}.grad.", r["micro"][m]["grads"], 3e-5, what="oracle twin micro-gradients")
        assert abs(r["norm"] - float(g[pre + "norm"])) < 1e-5 * float(g[pre + "norm"])
        # ---- the product: ONE call = accum_iter iterations = one optimiser step ----
        stats = A_fm.train_one_epoch(model, [b["labeled"] for b in batches], [b["unlabeled"] for b in batches], opt, dev,
                                     tw.epoch(s), tap, None, False, cfg)
        torch.cuda.synchronize()
        assert len(calls) == 2 * tw.accum and all(len(v) == tw.accum for v in grads.values())
        assert len(tap.norms) == 1, "the optimiser must step exactly once per accumulation window"
        assert abs(stats["lr"] - float(g[pre + "lr"])) < 1e-12
        for k in ("loss_total", "loss_x", "loss_u_s", "mask_ratio"):      # meter averages over the window's iterations
            assert abs(stats[k] - float(g[pre + k])) < TOL * max(abs(float(g[pre + k])), 1e-3), (k, stats[k], float(g[pre + k]))
        worst = 0.0
        for m in range(tw.accum):
            mp = f"{pre}m{m}."
            assert g[mp + "margins"][0] > 1.4e-5 and g[mp + "margins"][1] > 1.4e-5
            pred, logits = calls[2 * m], calls[2 * m + 1]
            assert rel(pred, g[mp + "pred_u_w"]) < TOL and rel(logits, g[mp + "logits"]) < TOL, f"step {s} micro-step {m}: logits"
            conf, mask, _ = SF.pseudo_label(pred)
            assert torch.equal(mask.cpu(), torch.from_numpy(g[mp + "mask"].astype(np.int64)))
            assert np.array_equal((conf >= cfg["conf_thresh"]).cpu().numpy(), g[mp + "keep"])
            worst = max(worst, check_rows(g, mp + "grad.", {k: v[m] for k, v in grads.items()}, TOL,
                                          what=f"step {s} micro-step {m} gradients (of loss / accum_iter)"))
        norm = float(torch.as_tensor(tap.norms[0]).item())
        assert abs(norm - float(g[pre + "norm"])) < TOL * float(g[pre + "norm"]), (norm, float(g[pre + "norm"]))
        assert norm > 1.9 * cfg["max_norm"]                                # pre-clip norm: the clip scaled the gradient by ~1/2
        sd = model.state_dict()
        check_packed(g, pre + "buf.", {k: v for k, v in sd.items() if "running" in k or "num_batches" in k}, 1e-5, what="buffers")
        for p in model.parameters():
            assert p.grad is None or float(p.grad.abs().max()) == 0.0      # zero_grad after the step
        # ---- the AdamW update of the CLIPPED accumulated gradient, element by element ----
        after = {k: p.detach().clone() for k, p in model.named_parameters()}
        lr = float(g[pre + "lr"])
        res = check_update_elementwise(before, after, {k: tw.o[k] for k in tw.pnames}, adamw_cond(tw.opt, tw.pnames), lr,
                                       what=f"step {s} AdamW after accumulation + clipping")
        ref_direct, cond_direct = {}, {}
        coef = [float(g[f"step{t}.norm"]) for t in range(s + 1)]
        coef = [min(1.0, cfg["max_norm"] / (n + 1e-6)) for n in coef]
        for k in tw.pnames:
            fk = pre + "upd.full." + k
            if fk in g.files and all((f"step{t}.m{m}.grad.full." + k) in g.files for t in range(s + 1) for m in range(tw.accum)):
                gt = [sum(torch.from_numpy(g[f"step{t}.m{m}.grad.full." + k]).double() for m in range(tw.accum)) * coef[t]
                      for t in range(s + 1)]
                v = gt[0] * gt[0] * 0.001
                for t in range(1, s + 1):
                    v = 0.999 * v + 0.001 * gt[t] * gt[t]
                cond_direct[k] = (v / (1.0 - 0.999 ** (s + 1))).sqrt()
                ref_direct[k] = before[k].double().cpu() + torch.from_numpy(g[fk]).double()
        assert len(ref_direct) >= 40
        check_update_elementwise(before, after, ref_direct, cond_direct, lr, what=f"step {s} AdamW vs the stored update", max_ill=1.0)
        print(f"{NAME} step {s}: worst micro-gradient statistic {worst:.2e}; norm {norm:.6f} (reference {float(g[pre + 'norm']):.6f}, "
              f"max_norm {cfg['max_norm']}); AdamW update worst deviation {res['worst_ratio']:.2f} of its bar, "
              f"{res['ill_frac']:.3%} ill-conditioned elements")
