"""GPU: TWO consecutive steps of every plugin's ``train_one_epoch`` (base / fixmatch / mean_teacher / cps / stpp) against the
reference's, on fixtures whose batches were searched to be free of ReLU / max-pool / arg-max / threshold near-ties at BOTH
steps (tools/make_golden.py::gen_step_case).  On such a state every correct fp32 implementation takes the same branches,
so nothing here tolerates a flipped decision:

* logits, teacher logits, losses <= 1e-4; arg-max / keep masks bit-exact; BN running statistics <= 1e-5;
* ALL 65 parameter gradients of each trainable model <= 1e-4 (row norms, row sums, random projections, full small tensors)
  - soft-target CE (MeanTeacher), in-module CE (base), the two CPS models, ST++;
* the AdamW UPDATE (after - before) element by element: <= (2e-4 + 1e-4 * rms(sqrt(v_hat)) / sqrt(v_hat)) of one lr-sized step
  (helpers.check_update_elementwise: an update error no larger than a 1e-4 gradient error can cause); the sign-like elements
  (sqrt(v_hat) <= 1e-6) are counted and bounded; same for the EMA teacher at (1 - decay) scale;
  against the reference directly for the tensors the fixture stores in full, against the live oracle twin (bit-identical
  to the reference in the build container, re-checked on this host first) for all 4 M elements.

Step 1 starts from the reference's post-step-0 state (weights, BN buffers, AdamW moments, teacher), loaded into the
product objects from the oracle twin: the fixture's step-1 batch is tie-free for THAT state.  The product continuing from
its own state is covered by tests/test_golden_gpu.py (oracle re-anchored on the device state) and tests/test_resume_gpu.py.
"""
import numpy as np
import pytest
import torch

from helpers import (StepfixTwin, adamw_cond, build_hip_model, check_packed, check_rows, check_update_elementwise, golden,
                     rel, to_dev)
from ssecg import functional as SF

pytestmark = pytest.mark.gpu
TOL = 1e-4
STEPFIX = ["stepfix_base_c1_b4_L250", "stepfix_fixmatch_c12_b2_L250", "stepfix_mean_teacher_c2_b2_L250",
           "stepfix_cps_c2_b1_L250", "stepfix_stpp_c12_b2_L250"]


class _Capture:
    def __init__(self, model, trainable=True):
        self.calls, self.grads = [], {}
        model.register_forward_hook(lambda m, i, o: self.calls.append(o["seg_logits"].detach().clone()))
        if trainable:
            for k, p in model.named_parameters():
                p.register_hook(lambda gr, k=k: self.grads.__setitem__(k, gr.detach().clone()))

    def clear(self):
        self.calls.clear(); self.grads.clear()


def _load_from_oracle(model, o_sd, optimizer=None, o_opt=None):
    """Product model (+ AdamW moments) <- oracle state, in place (pointer tables stay valid)."""
    sd = model.state_dict()
    with torch.no_grad():
        for k, v in sd.items():
            v.copy_(o_sd[k].detach().to(v.device))
    if optimizer is not None:
        for k, p in model.named_parameters():
            st = optimizer.state[p]
            assert int(torch.as_tensor(st["step"]).item()) == o_opt["step"]
            st["exp_avg"].copy_(o_opt["exp_avg." + k].to(p.device))
            st["exp_avg_sq"].copy_(o_opt["exp_avg_sq." + k].to(p.device))


def _params(model):
    return {k: p.detach().clone() for k, p in model.named_parameters()}


# Round 4: the BASELINE length for the two BASELINE plugins whose backward was pinned at L = 250 only (configs #3 and #1): one
# step of mean_teacher (C = 2, B = 1) and base (C = 1, B = 2) at L = 2000, the Winograd and the direct kernel selections; their
# batches were searched with the margin of gradfix_c12_b1_L2000 (5e-6: 2.8 M ReLU decisions per window pair leave no batch
# with 1.5e-5 in reach), which the F(4,3) kernels' single-element error (~4e-6 of the RMS) does not cross on these seeds.
STEPFIX_L2000 = ["stepfix_mean_teacher_c2_b1_L2000", "stepfix_base_c1_b2_L2000"]
CASES = ([pytest.param(n, True, id=n) for n in STEPFIX]
         + [pytest.param(n, w, id=f"{n}-{'winograd' if w else 'direct'}") for n in STEPFIX_L2000 for w in (True, False)])


@pytest.mark.parametrize("name,wino", CASES)
def test_two_steps_match_reference_on_tie_free_fixtures(name, wino, dev, monkeypatch):
    import algorithms.base as A_base
    import algorithms.cps as A_cps
    import algorithms.fixmatch as A_fm
    import algorithms.mean_teacher as A_mt
    import algorithms.stpp as A_stpp
    from utils.misc import NativeScalerWithGradNormCount
    from utils.optimizer import get_optimizer_from_config
    from ssecg import ops
    monkeypatch.setattr(ops, "WINOGRAD", wino)
    g = golden(name)
    tw = StepfixTwin(g)
    algo, B = tw.algo, tw.B
    mA = build_hip_model(tw.C, tw.sdA_np, dev)
    mB = build_hip_model(tw.C, tw.sdB_np, dev) if algo != "base" and algo != "fixmatch" else None
    if algo in ("mean_teacher", "stpp"):
        for p in mB.parameters():
            p.requires_grad = False
    if algo == "mean_teacher":
        with torch.no_grad():
            for pq, pk in zip(mA.parameters(), mB.parameters()):
                pk.data = pq.data                       # src/algorithms/mean_teacher.py:285-290 (Q4)
    cfg = dict(tw.cfg)
    optA = get_optimizer_from_config(cfg, mA.parameters())
    optB = get_optimizer_from_config(cfg, mB.parameters()) if algo == "cps" else None
    scaler = NativeScalerWithGradNormCount()
    capA = _Capture(mA)
    capB = _Capture(mB, trainable=(algo == "cps")) if mB is not None else None
    for s in range(tw.nsteps):
        pre = f"step{s}."
        need = 4.9e-6 if name in STEPFIX_L2000 else 1.4e-5
        assert float(g[pre + "fp32_vs_fp64_rel_l2"]) <= 1e-5 and g[pre + "margins"][0] > need and g[pre + "margins"][1] > need
        if s > 0:                                        # continue from the REFERENCE's state (module docstring)
            _load_from_oracle(mA, tw.oA, optA, tw.optA)
            if mB is not None:
                _load_from_oracle(mB, tw.oB, optB, tw.optB if algo == "cps" else None)
        batch_np, dmA, dmB = tw.inputs(s)
        batch = to_dev(batch_np, dev)
        mA.decode_head.fixed_dropout_mask = torch.from_numpy(dmA).to(dev, torch.uint8)
        if mB is not None:
            mB.decode_head.fixed_dropout_mask = torch.from_numpy(dmB).to(dev, torch.uint8)
        capA.clear()
        if capB: capB.clear()
        beforeA, beforeB = _params(mA), (_params(mB) if mB is not None else None)
        # ---- the live oracle twin, re-checked against the reference's stored statistics on this host ----
        r = tw.step(s)
        # (bit-identical in the build container; another host CPU sums in a different order: same class as the reference's
        # own fp32-vs-fp64 deviation of 3e-6 stored in the fixture)
        check_rows(g, pre + "grad.", r["grads"], 3e-5, what="oracle twin gradients")
        # ---- the product ----
        epoch = tw.epoch(s)
        if algo == "base":
            stats = A_base.train_one_epoch(mA, [batch["labeled"]], optA, dev, epoch, scaler, None, False, cfg)
            (logits,) = capA.calls
        elif algo == "fixmatch":
            stats = A_fm.train_one_epoch(mA, [batch["labeled"]], [batch["unlabeled"]], optA, dev, epoch, scaler, None, False, cfg)
            pred, logits = capA.calls
        elif algo == "mean_teacher":
            stats = A_mt.train_one_epoch(mA, mB, [batch["labeled"]], [batch["unlabeled"]], optA, dev, epoch, scaler, None, False, cfg)
            (pred,), (logits,) = capB.calls, capA.calls
        elif algo == "cps":
            stats = A_cps.train_one_epoch(mA, mB, [batch["labeled"]], [batch["unlabeled"]], optA, optB, dev, epoch, scaler, None,
                                          False, cfg)
            (pred, logits), (pred2, logits2) = capA.calls, capB.calls
        else:
            mB.eval()
            stats = A_stpp.train_one_epoch(mA, mB, [batch["labeled"]], [batch["unlabeled"]], optA, dev, epoch, scaler, None, False, cfg)
            (pred,), (logits,) = capB.calls, capA.calls
        torch.cuda.synchronize()
        lr = float(g[pre + "lr"])
        assert abs(stats["lr"] - lr) < 1e-12
        # ---- forward quantities against the reference ----
        assert rel(logits, g[pre + "logits"]) < TOL, f"{algo} step {s}: student logits"
        for k in ("loss", "loss_total", "loss_x", "loss_u_s", "mask_ratio"):
            if (pre + k) in g.files:
                assert abs(stats[k] - float(g[pre + k])) < TOL * max(abs(float(g[pre + k])), 1e-3), (k, stats[k], float(g[pre + k]))
        if algo != "base":
            assert rel(pred, g[pre + "pred_u_w"]) < TOL
            conf, mask, _ = SF.pseudo_label(pred)
            if algo != "mean_teacher":                   # soft targets: no discrete decision
                assert torch.equal(mask.cpu(), torch.from_numpy(g[pre + "mask"].astype(np.int64)))
            if algo == "fixmatch":
                assert np.array_equal((conf >= cfg["conf_thresh"]).cpu().numpy(), g[pre + "keep"])
        if algo == "cps":
            assert rel(pred2, g[pre + "pred_u_w_2"]) < TOL and rel(logits2, g[pre + "logits_2"]) < TOL
            assert torch.equal(SF.pseudo_label(pred2)[1].cpu(), torch.from_numpy(g[pre + "mask_2"].astype(np.int64)))
        # ---- gradients: every tensor at 1e-4, no flip tolerance ----
        wg = check_rows(g, pre + "grad.", dict(capA.grads), TOL, what=f"{algo} step {s} gradients")
        if algo == "cps":
            wg = max(wg, check_rows(g, pre + "grad2.", dict(capB.grads), TOL, what=f"{algo} step {s} gradients (model 2)"))
        # ---- BN buffers ----
        sdA = mA.state_dict()
        check_packed(g, pre + "buf.", {k: v for k, v in sdA.items() if "running" in k or "num_batches" in k}, 1e-5, what="buffers")
        if algo == "cps":
            sdB = mB.state_dict()
            check_packed(g, pre + "buf2.", {k: v for k, v in sdB.items() if "running" in k or "num_batches" in k}, 1e-5, what="buffers 2")
        # ---- the AdamW update, element by element ----
        afterA = _params(mA)
        res = check_update_elementwise(beforeA, afterA, {k: tw.oA[k] for k in tw.pnames}, adamw_cond(tw.optA, tw.pnames), lr,
                                       what=f"{algo} step {s} AdamW")
        # ... and against the reference itself for the tensors the fixture stores in full (update AND gradients of both steps)
        ref_direct, cond_direct = {}, {}
        for k in tw.pnames:
            fk = pre + "upd.full." + k
            if fk in g.files and ("step0.grad.full." + k) in g.files:
                g0 = torch.from_numpy(g["step0.grad.full." + k]).double()
                if s == 0:
                    cond_direct[k] = g0.abs()
                else:
                    g1 = torch.from_numpy(g["step1.grad.full." + k]).double()
                    cond_direct[k] = ((0.999 * 0.001 * g0 * g0 + 0.001 * g1 * g1) / (1.0 - 0.999 ** 2)).sqrt()
                ref_direct[k] = beforeA[k].double().cpu() + torch.from_numpy(g[fk]).double()
        assert len(ref_direct) >= 40
        check_update_elementwise(beforeA, afterA, ref_direct, cond_direct, lr, what=f"{algo} step {s} AdamW vs the stored update",
                                 max_ill=1.0)
        n_direct = sum(v.numel() for v in ref_direct.values())
        if algo == "cps":
            res2 = check_update_elementwise(beforeB, _params(mB), {k: tw.oB[k] for k in tw.pnames}, adamw_cond(tw.optB, tw.pnames),
                                            lr, what=f"{algo} step {s} AdamW (model 2)")
            res["ill_frac"] = max(res["ill_frac"], res2["ill_frac"])
        # ---- the EMA teacher ----
        if algo == "mean_teacher":
            afterB = _params(mB)
            # step 0: the teacher's parameters ARE the student's storage (Q4), so its first "EMA" equals the student's whole
            # AdamW step; from step 1 on it moves by (1 - decay) of the difference
            aliased = (s == 0)
            res_t = check_update_elementwise(beforeB, afterB, {k: tw.oB[k] for k in tw.pnames}, adamw_cond(tw.optA, tw.pnames), lr,
                                             what=f"teacher EMA step {s}", scale=1.0 if aliased else 1.0 - cfg["ema_decay"])
            tsd = mB.state_dict()
            check_packed(g, pre + "tbuf.", {k: v for k, v in tsd.items() if "running" in k or "num_batches" in k}, 1e-5,
                         what="teacher buffers")
            assert str(tsd["backbone.stem.1.num_batches_tracked"].dtype) == str(g[pre + "tbuf.nbt_dtype"])          # Q5
            print(f"  teacher EMA: worst deviation {res_t['worst_ratio']:.2f} of its bar, {res_t['worst_solid_lr']:.2e} (1-d) lr where solid")
        print(f"{name} step {s}: worst gradient statistic {wg:.2e}; AdamW update: {res['ill_frac']:.3%} ill-conditioned elements, "
              f"worst deviation {res['worst_ratio']:.2f} of its bar, {res['worst_solid_lr']:.2e} lr where sqrt(v_hat) > 1e-4; "
              f"{n_direct} elements checked against the reference's stored update")
