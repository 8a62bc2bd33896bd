"""CPU: the oracle (oracle/torch_ref.py) re-checked against the reference outputs frozen in tests/golden/
(the reference itself never leaves the build container).  The oracle is the checker used by every GPU test,
bench.py's cpu_baseline and smoke(); this file is what pins it."""
from collections import OrderedDict

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import DROPOUT_P, L, TRAIN_CFG, check_packed, dropout_mask_np, golden, sharpen_for
from oracle import torch_ref as O
from ssecg import synth


def _t(batch):
    return {g: {k: torch.from_numpy(v) for k, v in d.items()} for g, d in batch.items()}


@pytest.mark.parametrize("C,B,seed", [(1, 2, 11), (2, 2, 12), (12, 2, 13)])
def test_oracle_forward_matches_reference(C, B, seed):
    g = golden(f"forward_c{C}_b{B}")
    sd = O.state_from_numpy(synth.model_state(seed, C, trained=True, sharpen=sharpen_for(C)))
    x = torch.from_numpy(synth.normal(seed + 1, 1, (B, C, L)))
    with torch.no_grad():
        logits = O.model_forward(sd, x, train=False)
        conf, mask = O.pseudo_label(logits)
    assert (logits - torch.from_numpy(g["eval.logits"])).abs().max().item() < 1e-5
    assert np.array_equal(mask.numpy().astype(np.int8), g["eval.mask"])
    assert np.array_equal((conf >= TRAIN_CFG["conf_thresh"]).numpy(), g["eval.keep"])
    y = torch.from_numpy(synth.labels(seed + 1, 4, B, L))
    dm = torch.from_numpy(dropout_mask_np(seed + 1, B).astype(np.float32))
    lt = O.model_forward(sd, x, train=True, dropout_mask=dm, dropout_p=DROPOUT_P)
    assert (lt.detach() - torch.from_numpy(g["train.logits"])).abs().max().item() < 1e-5
    loss = F.cross_entropy(lt, y)
    assert abs(loss.item() - float(g["train.loss"])) < 1e-6
    names = O.param_names(sd)
    grads = dict(zip(names, torch.autograd.grad(loss, [sd[k] for k in names])))
    check_packed(g, "train.grad.", grads, 1e-5, what="oracle gradients")
    check_packed(g, "train.buf.", {k: sd[k] for k in O.buffer_names(sd)}, 1e-6, what="oracle buffers")


@pytest.mark.parametrize("algo,C,B,seed", [("fixmatch", 1, 2, 21), ("mean_teacher", 2, 2, 23), ("base", 1, 2, 24)])
def test_oracle_steps_match_reference(algo, C, B, seed):
    g = golden(f"{algo}_c{C}_b{B}")
    sd = O.state_from_numpy(synth.model_state(seed, C, trained=True, sharpen=sharpen_for(C)))
    teacher = None
    if algo == "mean_teacher":  # teacher parameters alias the student's at construction (Q4); own buffers
        tb = O.state_from_numpy(synth.model_state(seed + 50, C, trained=True, sharpen=sharpen_for(C)), requires_grad=False)
        pn = set(O.param_names(sd))
        teacher = OrderedDict((k, sd[k] if k in pn else tb[k]) for k in sd)
    opt, cfg = {}, dict(TRAIN_CFG, betas=(0.9, 0.999))
    for s in range(2):
        epoch = 3 + 9 * s
        batch = _t(synth.fixmatch_batch(seed + 10 + s, B, C, L))
        pre = f"step{s}."
        n = B if algo == "base" else 2 * B
        dm = torch.from_numpy(dropout_mask_np(seed + 10 + s, n).astype(np.float32))
        if algo == "base":
            r = O.supervised_step(sd, opt, batch["labeled"], cfg, epoch, dm)
            assert abs(r["loss"] - float(g[pre + "loss"])) < 1e-6
        elif algo == "fixmatch":
            r = O.fixmatch_step(sd, opt, batch, cfg, epoch, dm)
            assert np.array_equal(r["mask"].numpy().astype(np.int8), g[pre + "mask"])
            assert np.array_equal(r["keep"].numpy(), g[pre + "keep"])
            assert abs(r["mask_ratio"] - float(g[pre + "mask_ratio"])) < 1e-6
        else:
            r = O.mean_teacher_step(sd, teacher, opt, batch, cfg, epoch, dm)
        if algo != "base":
            for k in ("loss_total", "loss_x", "loss_u_s"):
                assert abs(r[k] - float(g[pre + k])) < 1e-6
        assert abs(r["lr"] - float(g[pre + "lr"])) < 1e-15
        assert (r["logits"] - torch.from_numpy(g[pre + "logits"])).abs().max().item() < 2e-5
        check_packed(g, pre + "param.", {k: sd[k] for k in O.param_names(sd)}, 1e-6, what="oracle params")
        if teacher is not None:
            check_packed(g, pre + "tparam.", {k: teacher[k] for k in O.param_names(sd)}, 1e-6, what="oracle teacher")
            assert str(teacher["backbone.stem.1.num_batches_tracked"].dtype) == str(g[pre + "tbuf.nbt_dtype"])


@pytest.mark.parametrize("algo,C,B,seed", [("cps", 2, 2, 25), ("stpp", 12, 2, 26)])
def test_oracle_pair_steps_match_reference(algo, C, B, seed):
    """CPS (two trainable models) and ST++ (student + frozen teacher) against the reference's train_one_epoch."""
    g = golden(f"{algo}_c{C}_b{B}")
    sdA = O.state_from_numpy(synth.model_state(seed, C, trained=True, sharpen=sharpen_for(C)))
    sdB = O.state_from_numpy(synth.model_state(seed + 50, C, trained=True, sharpen=sharpen_for(C)), requires_grad=(algo == "cps"))
    optA, optB, cfg = {}, {}, dict(TRAIN_CFG, betas=(0.9, 0.999))
    for s in range(2):
        epoch = 3 + 9 * s
        batch = _t(synth.fixmatch_batch(seed + 10 + s, B, C, L))
        pre = f"step{s}."
        dmA = torch.from_numpy(dropout_mask_np(seed + 10 + s, 2 * B).astype(np.float32))
        dmB = torch.from_numpy(dropout_mask_np(seed + 60 + s, 2 * B).astype(np.float32))
        if algo == "cps":
            r = O.cps_step(sdA, sdB, optA, optB, batch, cfg, epoch, (dmA, dmB))
            for i in (1, 2):
                assert np.array_equal(r[f"mask_{i}"].numpy().astype(np.int8), g[pre + f"mask_{i}"])
                assert (r[f"m{i}"]["logits"] - torch.from_numpy(g[pre + f"logits_{i}"])).abs().max().item() < 2e-5
            check_packed(g, pre + "param2.", {k: sdB[k] for k in O.param_names(sdB)}, 1e-6, what="oracle params (model 2)")
        else:
            r = O.stpp_step(sdA, sdB, optA, batch, cfg, epoch, dmA)
            assert np.array_equal(r["mask"].numpy().astype(np.int8), g[pre + "mask"])
            assert (r["logits"] - torch.from_numpy(g[pre + "logits"])).abs().max().item() < 2e-5
        for k in ("loss_total", "loss_x", "loss_u_s"):
            assert abs(r[k] - float(g[pre + k])) < 1e-6
        assert abs(r["lr"] - float(g[pre + "lr"])) < 1e-15
        check_packed(g, pre + "param.", {k: sdA[k] for k in O.param_names(sdA)}, 1e-6, what="oracle params")


def test_metrics_restatement_matches_reference_and_known_answers():
    from oracle import metrics_ref as M
    g = golden("stpp_select")
    K = 4
    preds = [p.astype(np.int64) for p in g["pred"]]
    rel = M.reliabilities(preds, K)
    assert np.allclose(rel, g["mious"].mean(axis=1), rtol=0, atol=1e-15)
    ref = M.select_reliable_ids(rel, num_models=len(preds), reference_ids=True)
    assert ref[0] == g["reference_reliable_ids"].tolist() and ref[1] == g["reference_unreliable_ids"].tolist()
    good, bad = M.select_reliable_ids(rel)
    assert sorted(good + bad) == list(range(len(rel))) and min(rel[good]) >= max(rel[bad])
    a, b = g["cm.a"], g["cm.b"]
    got = [M.calculate_miou(a, b), M.calculate_miou(a, b, True), M.calculate_miou(a[:1], b[:1])]
    assert np.allclose(got, g["cm.miou"], rtol=0, atol=1e-15)
    # torchmetrics-1.5.2 MeanIoU (published algorithm; the package is absent -> known-answer check only):
    # record 0: pred 0011, target 0101 -> IoU(c0) = 1/3, IoU(c1) = 1/3, c2 empty -> 0;  record 1: perfect on c2 only
    m = M.MeanIoURef(3)
    m.update(np.array([[0, 0, 1, 1], [2, 2, 2, 2]]), np.array([[0, 1, 0, 1], [2, 2, 2, 2]]))
    assert abs(m.compute() - ((1 / 3 + 1 / 3 + 0) / 3 + (0 + 0 + 1) / 3) / 2) < 1e-15
    m.update(np.array([[1, 1, 1, 1]]), np.array([[1, 1, 1, 1]]))   # second batch: score 1/3 -> mean over BATCHES
    assert abs(m.compute() - (((2 / 9) + (1 / 3)) / 2 + 1 / 3) / 2) < 1e-15
    mb = M.MeanIoURef(3, include_background=False, per_class=True)
    mb.update(np.array([[0, 0, 1, 1], [2, 2, 2, 2]]), np.array([[0, 1, 0, 1], [2, 2, 2, 2]]))
    assert np.allclose(mb.compute(), [(1 / 3 + 0) / 2, (0 + 1) / 2])


@pytest.mark.parametrize("name", ["gradfix_c12_b4_L250", "gradfix_c1_b2_L500", "gradfix_c12_b1_L2000"])
def test_oracle_matches_tie_free_gradient_fixtures(name):
    """The searched well-conditioned fixtures (tools/make_golden.py::gen_gradient_case): on them every correct fp32
    implementation takes the same ReLU / max-pool / arg-max branches, so the oracle must hit the reference's gradients at
    1e-5 on ANY host CPU (row norms, random projections, full small tensors) - the same checks the GPU test applies to
    the HIP path at 1e-4."""
    g = golden(name)
    C, B, Lg, seed, bseed, feat_len = (int(v) for v in g["meta"])
    assert float(g["fp32_vs_fp64_rel_l2"]) <= 1e-5 and g["margins"][0] > 5e-6 and g["margins"][1] > 5e-6
    sd = O.state_from_numpy(synth.model_state(seed, C, trained=True, sharpen=1.0))
    cfg = dict(TRAIN_CFG, betas=(0.9, 0.999), conf_thresh=float(g["conf_thresh"]))
    dm = torch.from_numpy(dropout_mask_np(bseed, 2 * B, lp=feat_len).astype(np.float32))
    r = O.fixmatch_step(sd, {}, _t(synth.fixmatch_batch(bseed, B, C, Lg)), cfg, 3, dm)
    assert np.array_equal(r["mask"].numpy().astype(np.int8), g["mask"]) and np.array_equal(r["keep"].numpy(), g["keep"])
    assert (r["logits"] - torch.from_numpy(g["logits"])).abs().max().item() < 2e-5
    for k in ("loss_total", "loss_x", "loss_u_s", "mask_ratio"):
        assert abs(r[k] - float(g[k])) < 1e-6
    for i, k in enumerate(str(n) for n in g["grad.names"]):
        got = r["grads"][k].double()
        rows = got.reshape(got.shape[0], -1) if got.dim() > 1 else got.reshape(1, -1)
        ref_l2 = g["grad.rowl2." + k]
        scale = float(np.sqrt((ref_l2 ** 2).mean())) + 1e-300
        assert np.abs(rows.pow(2).sum(dim=1).sqrt().numpy() - ref_l2).max() / scale < 1e-5, k
        if ("grad.full." + k) in g.files:
            ref_t = torch.from_numpy(g["grad.full." + k]).double()
            assert ((got - ref_t).norm() / (ref_t.norm() + 1e-300)).item() < 1e-5, k


@pytest.mark.parametrize("name", ["stepfix_base_c1_b4_L250", "stepfix_fixmatch_c12_b2_L250", "stepfix_mean_teacher_c2_b2_L250",
                                  "stepfix_cps_c2_b1_L250", "stepfix_stpp_c12_b2_L250",
                                  "stepfix_mean_teacher_c2_b1_L2000", "stepfix_base_c1_b2_L2000"])
def test_oracle_matches_two_step_tie_free_fixtures(name):
    """Every plugin's two reference steps (tools/make_golden.py::gen_step_case; both batches searched tie-free): the oracle
    twin used by tests/test_stepfix_gpu.py must reproduce the reference's logits, losses, masks, ALL gradients (1e-5) and
    the AdamW / EMA UPDATES (row statistics of after - before at 1e-5 of an lr-sized step, full small tensors) on any host."""
    from helpers import StepfixTwin, check_rows
    g = golden(name)
    tw = StepfixTwin(g)
    for s in range(tw.nsteps):
        pre = f"step{s}."
        assert float(g[pre + "fp32_vs_fp64_rel_l2"]) <= 1e-5
        beforeA = {k: tw.oA[k].detach().clone() for k in tw.pnames}
        beforeB = {k: tw.oB[k].detach().clone() for k in tw.pnames} if tw.oB is not None else None
        r = tw.step(s)
        assert abs(r["lr"] - float(g[pre + "lr"])) < 1e-15
        assert (r["logits"] - torch.from_numpy(g[pre + "logits"])).abs().max().item() < 2e-5
        for k in ("loss_total", "loss_x", "loss_u_s", "mask_ratio"):
            if (pre + k) in g.files and k in r:
                assert abs(r[k] - float(g[pre + k])) < 1e-6, k
        if tw.algo == "base":
            assert abs(r["loss"] - float(g[pre + "loss"])) < 1e-6
        if "mask" in r:
            assert np.array_equal(r["mask"].numpy().astype(np.int8), g[pre + "mask"])
        if "keep" in r:
            assert np.array_equal(r["keep"].numpy(), g[pre + "keep"])
        check_rows(g, pre + "grad.", r["grads"], 1e-5, what=f"{name} step {s} oracle gradients")
        check_rows(g, pre + "upd.", {k: tw.oA[k].detach().double() - beforeA[k].double() for k in tw.pnames}, 1e-5,
                   what=f"{name} step {s} oracle AdamW update")
        check_packed(g, pre + "buf.", {k: tw.oA[k] for k in O.buffer_names(tw.oA)}, 1e-6, what="oracle buffers")
        if tw.algo == "cps":
            assert (r["logits_2"] - torch.from_numpy(g[pre + "logits_2"])).abs().max().item() < 2e-5
            assert np.array_equal(r["mask_2"].numpy().astype(np.int8), g[pre + "mask_2"])
            check_rows(g, pre + "grad2.", r["grads2"], 1e-5, what="oracle gradients (model 2)")
            check_rows(g, pre + "upd2.", {k: tw.oB[k].detach().double() - beforeB[k].double() for k in tw.pnames}, 1e-5,
                       what="oracle AdamW update (model 2)")
        if tw.algo == "mean_teacher":
            check_rows(g, pre + "tupd.", {k: tw.oB[k].detach().double() - beforeB[k].double() for k in tw.pnames}, 1e-5,
                       what="oracle EMA update")
            check_packed(g, pre + "tbuf.", {k: tw.oB[k] for k in O.buffer_names(tw.oB)}, 1e-6, what="oracle teacher buffers")
            assert str(tw.oB["backbone.stem.1.num_batches_tracked"].dtype) == str(g[pre + "tbuf.nbt_dtype"])


def test_oracle_matches_the_accumulation_and_clipping_fixture():
    """tools/make_golden.py::gen_accum_case: the reference's real FixMatch loop with accum_iter = 2 and an active max_norm over
    two optimiser steps (four tie-free micro-batches).  oracle/torch_ref.fixmatch_accum_step - the twin of
    tests/test_accum_gpu.py - must reproduce every micro-step's logits / masks / gradients (of loss / accum_iter), the
    pre-clip norm the scaler returns, and the AdamW update of the clipped accumulated gradient on any host."""
    import importlib.util, os
    from helpers import check_rows
    spec = importlib.util.spec_from_file_location("test_accum_gpu", os.path.join(os.path.dirname(__file__), "test_accum_gpu.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    g = golden(mod.NAME)
    tw = mod.AccumTwin(g)
    for s in range(tw.nsteps):
        pre = f"step{s}."
        before = {k: tw.o[k].detach().clone() for k in tw.pnames}
        r = tw.step(s)
        assert abs(r["lr"] - float(g[pre + "lr"])) < 1e-15
        assert abs(r["norm"] - float(g[pre + "norm"])) < 1e-5 * float(g[pre + "norm"]) and r["norm"] > 1.9 * tw.cfg["max_norm"]
        for m in range(tw.accum):
            mp = f"{pre}m{m}."
            assert (r["micro"][m]["logits"] - torch.from_numpy(g[mp + "logits"])).abs().max().item() < 2e-5
            assert np.array_equal(r["micro"][m]["mask"].numpy().astype(np.int8), g[mp + "mask"])
            assert np.array_equal(r["micro"][m]["keep"].numpy(), g[mp + "keep"])
            check_rows(g, mp + "grad.", r["micro"][m]["grads"], 1e-5, what=f"step {s} micro-step {m} oracle gradients")
        for k in ("loss_total", "loss_x", "loss_u_s", "mask_ratio"):      # the plugin reports the window's averages
            assert abs(sum(r["micro"][m][k] for m in range(tw.accum)) / tw.accum - float(g[pre + k])) < 1e-6, k
        check_rows(g, pre + "upd.", {k: tw.o[k].detach().double() - before[k].double() for k in tw.pnames}, 1e-5,
                   what=f"step {s} oracle AdamW update (clipped accumulated gradient)")
        check_packed(g, pre + "buf.", {k: tw.o[k] for k in O.buffer_names(tw.o)}, 1e-6, what="oracle buffers")


# ---- use_amp: oracle/amp_ref.py against the reference under PyTorch's CPU bf16 autocast (round 5) ---------
# tools/make_golden.py::gen_amp_blocks / gen_amp_case ran the reference's REAL train_one_epoch(use_amp=True) with
# torch.cuda.amp.autocast bound to torch.autocast("cpu", bfloat16).  amp_ref is an emulation (fp32 ops + explicit bf16
# roundings); these tests are what pins it - block by block where a 16-bit computation is not chaotic, and statistically on
# the whole two-step chain.  Policy "cpu_autocast" places the roundings where PyTorch's CPU autocast does; policy "hip" is what
# the HIP path implements (fp32 classifier tail / weight gradients - the documented deviations; "hip_fp32_stem" is the
# SSECG_AMP_STEM_LP=0 variant with the fp32 stem of rounds 2-4).
from helpers import AMP_BLOCKS, AmpfixCase, bf16_from_bits, rowl2_err, rows_l2  # noqa: E402


def _l2(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).norm() / (b.norm() + 1e-300))


@pytest.mark.parametrize("pol", ["cpu_autocast", "hip"])
def test_amp_emulation_blocks_against_reference_autocast(pol):
    from oracle import amp_ref as A
    g = golden("ampfix_blocks_c12_b2_L2000")
    C, B, Lg, seed, feat_len, bseed = (int(v) for v in g["meta"])
    sd = O.state_from_numpy(synth.model_state(seed, C, trained=True, sharpen=1.0))

    def params_err(prefix):
        ks = [k for k in sd if k.startswith(prefix + ".") and sd[k].requires_grad]
        e = (max(rows_l2(sd[k].grad, g["grad.rows." + k]) for k in ks), max(rowl2_err(sd[k].grad, g["grad.rowl2." + k]) for k in ks))
        for k in ks:
            sd[k].grad = None
        return e

    with A.policy(pol):
        # the eight BasicBlocks and the head's conv unit are 16-bit under BOTH policies: isolated 1-ulp flips only
        for tag, tin, prefix, stride, has_ds in AMP_BLOCKS:
            xin = bf16_from_bits(g["act." + tin]).requires_grad_(True)
            out = A._basic_block(sd, prefix, xin, stride, has_ds)
            out.backward(bf16_from_bits(g["gact." + tag]))
            ref = bf16_from_bits(g["act." + tag])
            mm = float((out.detach() != ref).float().mean())
            e_rows, e_norm = params_err(prefix)
            assert _l2(out, ref) < 1e-3 and mm < 1e-2, (tag, _l2(out, ref), mm)
            assert _l2(xin.grad, bf16_from_bits(g["gact." + tin])) < 1e-2, tag
            assert e_rows < 1e-2 and e_norm < 1e-2, (tag, e_rows, e_norm)
        h = bf16_from_bits(g["act.layer4.1"]).requires_grad_(True)
        a = A.head_unit(sd, h)
        a.backward(bf16_from_bits(g["gact.headconv"]))
        assert _l2(a, bf16_from_bits(g["act.headconv"])) < 1e-3
        assert _l2(h.grad, bf16_from_bits(g["gact.layer4.1"])) < 1e-2
        assert max(params_err("decode_head.convs.0")) < 1e-2
        # classifier tail: bf16 under autocast (bit-exact low-resolution logits), fp32 under "hip"; stem: 16-bit under both
        tight = pol == "cpu_autocast"       # (the classifier tail; the stem is 16-bit under both policies since round 5)
        batch = synth.learnable_batch(bseed, B, C, Lg)
        pooled = A.stem_forward(sd, torch.from_numpy(batch["labeled"]["ecg"]))
        pooled.backward(bf16_from_bits(g["gact.pool"]))
        k = "backbone.stem.0.weight"
        assert _l2(pooled, bf16_from_bits(g["act.pool"])) < 1e-4
        assert _l2(sd[k].grad, torch.from_numpy(g["grad.full." + k])) < (1e-3 if tight else 5e-3)   # hip: weight gradient not rounded to bf16
        a2 = bf16_from_bits(g["act.headconv"]).requires_grad_(True)
        dm = torch.from_numpy(dropout_mask_np(bseed, B, lp=feat_len).astype(np.float32))
        lo, logits = A.head_tail(sd, a2, Lg, dm)
        loss = F.cross_entropy(logits, torch.from_numpy(batch["labeled"]["target"]))
        loss.backward()
        assert _l2(lo, bf16_from_bits(g["act.lowres"])) < (1e-4 if tight else 1e-2)
        assert _l2(logits, bf16_from_bits(g["act.logits"])) < 1e-2      # bf16 interpolation kernel: one-ulp differences in many elements
        assert abs(loss.item() - float(g["loss"])) < (5e-5 if tight else 1e-3) * float(g["loss"])
        assert _l2(a2.grad, bf16_from_bits(g["gact.headconv"])) < 1e-2
        assert _l2(sd["decode_head.cls_seg.weight"].grad, torch.from_numpy(g["grad.full.decode_head.cls_seg.weight"])) < 1e-2


@pytest.mark.parametrize("name", ["ampfix_fixmatch_c12_b16_L2000", "ampfix_mean_teacher_c2_b8_L2000", "ampfix_base_c1_b8_L2000",
                                  "ampfix_stpp_c12_b8_L2000", "ampfix_cps_c2_b8_L2000"])
def test_amp_emulation_steps_against_reference_autocast(name):
    """Two-step chains.  The fixture stores the distances measured at generation (emu_cpu / emu_hip); re-measured here they must
    agree with the stored yardsticks (same host: equal; another CPU sums in another order and a 16-bit chain amplifies that, hence
    the factors), the autocast-placed emulation must be CLOSER to the reference's vectors than the reference's own fp32 run is,
    and its step-0 losses within 1e-3."""
    g = golden(name)
    case = AmpfixCase(g)
    for pol, tag in (("cpu_autocast", "emu_cpu"), ("hip", "emu_hip")):
        res = case.emulate(pol)
        for s, r in enumerate(res):
            losses = {"loss": r.get("loss"), "loss_total": r.get("loss_total"), "loss_x": r.get("loss_x"), "loss_u_s": r.get("loss_u_s")}
            # CPS (src/algorithms/cps.py:96-157): both models' vectors - model 2's under "m2."
            views = [("", r["logits"], r["grads"])] if case.algo != "cps" else [("", r["logits_1"], r["grads_1"]), ("m2.", r["logits_2"], r["grads_2"])]
            for sub, logits, grads in views:
                pre = f"step{s}." + sub
                d = case.distances(s, logits, losses, grads, sub)
                st = {k: g[pre + tag + "." + k] for k in ("logits_l2", "loss_err", "rows_cos", "norm_err")}
                assert d["logits_l2"] <= 1.5 * float(st["logits_l2"]) + 1e-2, (pol, s, d["logits_l2"], float(st["logits_l2"]))
                assert (d["loss_err"] <= 1.5 * st["loss_err"].max() + 1e-3).all(), (pol, s, d["loss_err"], st["loss_err"])
                assert (d["rows_cos"] >= st["rows_cos"] - (0.05 + 2.0 * case.floor(s, sub))).all(), (pol, s)
                assert (d["norm_err"] <= 1.5 * st["norm_err"] + 0.05 + 2.0 * case.norm_floor(s, sub)).all(), (pol, s)
                if pol == "cpu_autocast" and s == 0:
                    assert d["logits_l2"] < float(g[pre + "fp32.logits_l2"])
                    assert d["rows_cos"].min() > g[pre + "fp32.rows_cos"].min()
                    assert d["loss_err"].max() < 1e-3


@pytest.mark.parametrize("name", ["ampfix_eval_c12_b4_L2000", "ampfix_eval_c1_b4_L2000"])
def test_amp_emulation_eval_against_reference_autocast(name):
    """``evaluate()`` under use_amp: the reference's eval-mode forward runs INSIDE autocast (src/algorithms/base.py:202).  The
    emulation's eval-mode placement against the reference's real ``evaluate(use_amp=True)`` under CPU bf16 autocast: block by block
    fed the reference's own bf16 input (5-7 roundings deep: isolated 1-ulp flips only), then the free-running pass - logits, the
    logged loss, the arg-max of the (16-bit under autocast) probabilities and the per-record confusion counts."""
    from oracle import amp_ref as A
    from helpers import AmpEvalCase
    g = golden(name)
    case = AmpEvalCase(g)
    sd = O.state_from_numpy(case.sd_np, requires_grad=False)
    batches = [{k: torch.from_numpy(v) for k, v in b.items()} for b in case.batches()]
    for pol, tag in (("cpu_autocast", "emu_cpu"), ("hip", "emu_hip")):
        with A.policy(pol):
            with torch.no_grad():
                pooled = A.stem_forward_eval(sd, batches[0]["ecg"])
                assert _l2(pooled, bf16_from_bits(g["act.pool"])) < 1e-4
                for t, tin, prefix, stride, has_ds in AMP_BLOCKS:
                    out = A._basic_block_eval(sd, prefix, bf16_from_bits(g["act." + tin]), stride, has_ds)
                    ref = bf16_from_bits(g["act." + t])
                    assert _l2(out, ref) < 1e-3 and float((out != ref).float().mean()) < 1e-2, (pol, t)
                a = A._unit_eval(sd, "decode_head.convs.0.0", "decode_head.convs.0.1", bf16_from_bits(g["act.layer4.1"]), 1, 1)
                assert _l2(a, bf16_from_bits(g["act.headconv"])) < 1e-3
                lo, _ = A.head_tail(sd, bf16_from_bits(g["act.headconv"]), case.L, None, 0.0)
                assert _l2(lo, bf16_from_bits(g["act.lowres"])) < (1e-4 if pol == "cpu_autocast" else 1e-2)
            rs = [A.evaluate_batch(sd, b) for b in batches]
        ref_logits = torch.cat([bf16_from_bits(l) for l in g["logits"]])
        el = torch.cat([r["logits"] for r in rs])
        assert _l2(el, ref_logits) <= 1.5 * float(g[tag + ".logits_l2"]) + 2e-3
        loss = float(np.mean([r["loss"] for r in rs]))
        assert abs(loss - float(g["loss"])) <= (1.5 * float(g[tag + ".loss_err"]) + 5e-4) * float(g["loss"])
        pred = torch.cat([r["pred"] for r in rs]).numpy().astype(np.int8)
        clear = case.clear()
        assert clear.mean() > 0.97
        assert np.array_equal(pred[clear], g["pred"][clear]), (pol, float((pred != g["pred"])[clear].mean()))
        assert float((pred != g["pred"]).mean()) <= 2.0 * float(g[tag + ".pred_mismatch"]) + 5e-4
    # the reference's own fp32 pass beside it (use_amp=False) is what the fp32 oracle reproduces
    with torch.no_grad():
        l32 = [O.model_forward(sd, b["ecg"], train=False) for b in batches]
    loss32 = float(np.mean([float(F.cross_entropy(l, b["target"])) for l, b in zip(l32, batches)]))
    assert abs(loss32 - float(g["fp32.loss"])) < 1e-5 * max(float(g["fp32.loss"]), 1.0)
    assert float((torch.cat(l32).argmax(dim=1).numpy().astype(np.int8) != g["fp32.pred"]).mean()) < 1e-4


@pytest.mark.parametrize("pol", ["cpu_autocast", "hip"])
def test_amp_emulation_backward_chains_across_stage_boundaries(pol):
    """The sharp counterpart of the two-step chain statistics: blocks chained through AUTOGRAD across each stage boundary (and the
    whole eight-block body) with the forward teacher-forced (helpers.ReplaceForward) - the gradient entering the chain is the
    reference's, every later gradient is what the previous block's backward produced, the two branch gradients at the stride-2
    block's input are summed by the chain itself.  Input gradient <= 1.5e-2 (three blocks) / 3e-2 (eight), parameter gradients of every
    block on the way <= 2.5e-2 / 4e-2 (measured 4.5e-3 - 1.2e-2 and <= 1.8e-2)."""
    from oracle import amp_ref as A
    from helpers import AMP_CHAINS, ReplaceForward
    g = golden("ampfix_blocks_c12_b2_L2000")
    C, B, Lg, seed, feat_len, bseed = (int(v) for v in g["meta"])
    order = [b[0] for b in AMP_BLOCKS]
    info = {b[0]: b for b in AMP_BLOCKS}
    for first, last in AMP_CHAINS:
        chain = order[order.index(first):order.index(last) + 1]
        sd = O.state_from_numpy(synth.model_state(seed, C, trained=True, sharpen=1.0))
        with A.policy(pol):
            xin = bf16_from_bits(g["act." + info[first][1]]).requires_grad_(True)
            h = xin
            for t in chain:
                _, _, prefix, stride, has_ds = info[t]
                h = A._basic_block(sd, prefix, h, stride, has_ds)
                if t != last:
                    h = ReplaceForward.apply(h, bf16_from_bits(g["act." + t]))
            h.backward(bf16_from_bits(g["gact." + last]))
        bar_in, bar_p = (1.5e-2, 2.5e-2) if len(chain) == 3 else (3e-2, 4e-2)
        assert _l2(xin.grad, bf16_from_bits(g["gact." + info[first][1]])) < bar_in, (first, last)
        for t in chain:
            ks = [k for k in sd if k.startswith(info[t][2] + ".") and sd[k].requires_grad]
            assert max(rows_l2(sd[k].grad, g["grad.rows." + k]) for k in ks) < bar_p, (first, last, t)
            assert max(rowl2_err(sd[k].grad, g["grad.rowl2." + k]) for k in ks) < bar_p, (first, last, t)
