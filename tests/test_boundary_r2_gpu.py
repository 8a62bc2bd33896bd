"""GPU: boundary pieces added in round 2 - the GradScaler bookkeeping / gradient norm / inf-skip kernels
(src/utils/misc.py:236-278), the fused SGD (src/utils/optimizer.py:15-26), the FCNHead constructor variants
(src/models/decode_heads/fcn_head.py:49-97), evaluate()'s fast path and test()'s artefacts
(src/algorithms/base.py:184-245,442-499)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import TRAIN_CFG, build_hip_model, model_cfg, rel
from ssecg import functional as SF
from ssecg import ops, synth
from ssecg.optim import FusedAdamW, FusedSGD

pytestmark = pytest.mark.gpu


def _params(dev, shapes=((64, 12, 7), (64,), (128, 64, 3), (4, 128, 1), (4,), (3000,)), seed=1):
    ps = []
    for i, s in enumerate(shapes):
        p = torch.nn.Parameter(torch.from_numpy(synth.normal(seed, 20 + i, s)).to(dev))
        p.grad = torch.from_numpy(synth.normal(seed, 60 + i, s, std=0.3)).to(dev)
        ps.append(p)
    return ps


def test_grad_norm_scaler_update_and_inf_skip(dev):
    from utils.misc import NativeScalerWithGradNormCount
    ps = _params(dev)
    opt = FusedAdamW(ps, lr=1e-3, weight_decay=0.05)
    ref_norm = torch.norm(torch.stack([torch.norm(p.grad.detach().double().cpu(), 2) for p in ps]), 2).item()
    out = opt.grad_norm()
    assert abs(out[0].item() - ref_norm) < 1e-6 * ref_norm and out[1].item() == 0.0
    # GradScaler.update(): growth after `growth_interval` clean steps, back-off + skipped update on a non-finite gradient
    sc = NativeScalerWithGradNormCount()
    sc._host["growth_interval"] = 3
    before = [p.detach().clone() for p in ps]
    for i in range(3):
        loss = sum((p * 0.0).sum() for p in ps)           # zero loss: .grad tensors stay as set (backward adds zeros)
        norm = sc(loss, opt, clip_grad=None, parameters=ps, update_grad=True)
    assert abs(norm.item() - ref_norm) < 1e-6 * ref_norm
    st = sc.state_dict()
    assert st == {"scale": 131072.0, "growth_factor": 2.0, "backoff_factor": 0.5, "growth_interval": 3, "_growth_tracker": 0}
    assert all(not torch.equal(a, p.detach()) for a, p in zip(before, ps))
    mid = [p.detach().clone() for p in ps]
    m_before = [opt.state[p]["exp_avg"].clone() for p in ps]
    ps[2].grad[5, 3, 1] = float("inf")
    loss = sum((p * 0.0).sum() for p in ps)
    norm = sc(loss, opt, clip_grad=None, parameters=ps, update_grad=True)
    assert not np.isfinite(norm.item())
    st = sc.state_dict()
    assert st["scale"] == 65536.0 and st["_growth_tracker"] == 0
    for a, p, m0 in zip(mid, ps, m_before):                 # the update was skipped inside the kernel
        assert torch.equal(a, p.detach()) and torch.equal(opt.state[p]["exp_avg"], m0)
    # torch's per-parameter step counters do not count the skipped update (GradScaler.step never called optimizer.step)
    assert int(opt.state_dict()["state"][0]["step"]) == 3
    ps[2].grad[5, 3, 1] = 0.25
    # a checkpointed scaler resumes
    sc2 = NativeScalerWithGradNormCount(); sc2.load_state_dict(st)
    sc2(sum((p * 0.0).sum() for p in ps), opt, parameters=ps)
    assert sc2.state_dict()["_growth_tracker"] == 1 and sc2.state_dict()["scale"] == 65536.0
    # clip_grad_norm_
    g0 = [p.grad.clone() for p in ps]
    n0 = opt.grad_norm()[0].item()
    out = opt.grad_norm(max_norm=0.5 * n0)
    coef = 0.5 * n0 / (n0 + 1e-6)
    for a, p in zip(g0, ps):
        assert rel(p.grad, a * coef) < 1e-6
    opt.grad_norm(max_norm=10 * n0)                          # norm below the bound: untouched
    for a, p in zip(g0, ps):
        assert rel(p.grad, a * coef) < 1e-6


def test_adamw_trajectory_after_a_skipped_step_matches_torch(dev):
    """An inf gradient at step k: GradScaler.step does not call optimizer.step, so torch's bias corrections 1 - beta^t keep
    counting REAL steps.  The fused path skips inside the kernel (no host read) and must produce the same updates at steps
    k+1..k+3 - the kernel subtracts the optimiser's own device-side skip counter from the launch count - and the same
    per-parameter ``step`` in the checkpoint.  Two optimisers sharing one scaler (CPS) keep separate counters."""
    from utils.misc import NativeScalerWithGradNormCount
    ps, ps2 = _params(dev, seed=5), _params(dev, seed=6)
    ref = [torch.nn.Parameter(p.detach().cpu().clone()) for p in ps]
    opt, opt2 = FusedAdamW(ps, lr=1e-3, weight_decay=0.05), FusedAdamW(ps2, lr=1e-3, weight_decay=0.05)
    ropt = torch.optim.AdamW(ref, lr=1e-3, weight_decay=0.05)
    sc = NativeScalerWithGradNormCount()
    real_steps = 0
    for step in range(6):
        for i, (p, r) in enumerate(zip(ps, ref)):
            g = torch.from_numpy(synth.normal(40 + step, 60 + i, tuple(p.shape), std=0.3))
            p.grad = g.to(dev); r.grad = g.clone()
        for i, p in enumerate(ps2):
            p.grad = torch.from_numpy(synth.normal(50 + step, 60 + i, tuple(p.shape), std=0.3)).to(dev)
        if step == 2:
            ps[1].grad.view(-1)[7] = float("nan")        # only optimiser 1's gradients are non-finite
        else:
            ropt.step(); real_steps += 1                 # torch + GradScaler: the step-2 update never happens
        sc(sum((p * 0.0).sum() for p in ps), opt, parameters=ps)
        sc(sum((p * 0.0).sum() for p in ps2), opt2, parameters=ps2)
        if step != 2:
            for p, r in zip(ps, ref):
                assert rel(p, r) < 2e-6, f"step {step}: parameters diverge from torch.optim.AdamW"
    sd, rsd = opt.state_dict(), ropt.state_dict()
    assert real_steps == 5
    for i in range(len(ps)):
        assert int(sd["state"][i]["step"]) == int(rsd["state"][i]["step"]) == 5
        assert rel(sd["state"][i]["exp_avg"], rsd["state"][i]["exp_avg"]) < 2e-6
        assert rel(sd["state"][i]["exp_avg_sq"], rsd["state"][i]["exp_avg_sq"]) < 2e-6
    assert int(opt2.state_dict()["state"][0]["step"]) == 6      # the other optimiser skipped nothing
    # after the checkpoint-time reconciliation the trajectory still follows torch
    for step in range(6, 8):
        for i, (p, r) in enumerate(zip(ps, ref)):
            g = torch.from_numpy(synth.normal(40 + step, 60 + i, tuple(p.shape), std=0.3))
            p.grad = g.to(dev); r.grad = g.clone()
        ropt.step()
        sc(sum((p * 0.0).sum() for p in ps), opt, parameters=ps)
        for p, r in zip(ps, ref):
            assert rel(p, r) < 2e-6
    assert int(opt.state_dict()["state"][0]["step"]) == 7


@pytest.mark.parametrize("momentum,wd", [(0.0, 0.0), (0.9, 0.0), (0.9, 0.05), (0.0, 0.05)])
def test_fused_sgd_matches_torch(momentum, wd, dev):
    ps = _params(dev, seed=3)
    ref = [torch.nn.Parameter(p.detach().cpu().clone()) for p in ps]
    opt = FusedSGD(ps, lr=0.1, momentum=momentum, weight_decay=wd)
    ropt = torch.optim.SGD(ref, lr=0.1, momentum=momentum, weight_decay=wd)
    for step in range(3):
        for i, (p, r) in enumerate(zip(ps, ref)):
            g = torch.from_numpy(synth.normal(30 + step, 60 + i, tuple(p.shape), std=0.3))
            p.grad = g.to(dev); r.grad = g.clone()
        opt.step(); ropt.step()
        for p, r in zip(ps, ref):
            assert rel(p, r) < 2e-6
    sd, rsd = opt.state_dict(), ropt.state_dict()
    assert set(sd["param_groups"][0]) >= {"lr", "momentum", "dampening", "weight_decay", "nesterov"}
    if momentum:
        for i in range(len(ps)):
            assert rel(sd["state"][i]["momentum_buffer"], rsd["state"][i]["momentum_buffer"]) < 2e-6
    from utils.optimizer import get_optimizer_from_config
    o = get_optimizer_from_config(dict(optimizer="sgd", lr=0.01, weight_decay=1e-4, optimizer_kwargs={"momentum": 0.9}), ps)
    assert isinstance(o, FusedSGD) and o.param_groups[0]["momentum"] == 0.9


def _ref_head(sd, feats, in_index, num_convs, concat_input, pad, dil, train, mask, p):
    x = feats[in_index]
    out = x
    for i in range(num_convs):
        out = F.conv1d(out, sd[f"convs.{i}.0.weight"], padding=pad, dilation=dil)
        out = F.relu(F.batch_norm(out, sd[f"convs.{i}.1.running_mean"].clone(), sd[f"convs.{i}.1.running_var"].clone(),
                                  sd[f"convs.{i}.1.weight"], sd[f"convs.{i}.1.bias"], training=train, momentum=0.1, eps=1e-5))
    if concat_input:
        out = F.conv1d(torch.cat([x, out], dim=1), sd["conv_cat.0.weight"], padding=1)
        out = F.relu(F.batch_norm(out, sd["conv_cat.1.running_mean"].clone(), sd["conv_cat.1.running_var"].clone(),
                                  sd["conv_cat.1.weight"], sd["conv_cat.1.bias"], training=train, momentum=0.1, eps=1e-5))
    if train and mask is not None:
        out = out * mask * (1.0 / (1.0 - p))
    return F.conv1d(out, sd["cls_seg.weight"], sd["cls_seg.bias"])


@pytest.mark.parametrize("num_convs,concat_input,in_ch", [(2, False, 64), (1, True, 64), (2, True, 64), (0, True, 32), (0, False, 32)])
def test_fcn_head_variants(num_convs, concat_input, in_ch, dev):
    """FCNHead(num_convs, concat_input) other than the shipped (1, False): forward (train + eval) and all parameter
    gradients against a torch-CPU restatement of fcn_head.py:89-97."""
    from models.decode_heads import FCNHead
    torch.manual_seed(5)
    ch, K, N, Lf, p = 32, 4, 3, 63, 0.1
    head = FCNHead(in_channels=in_ch, channels=ch, num_classes=K, num_convs=num_convs, concat_input=concat_input, dropout_ratio=p,
                   in_index=-1)
    with torch.no_grad():
        for name, t in head.named_parameters():
            if name.endswith("1.weight"): t.copy_(1.0 + 0.2 * torch.randn_like(t))
            if name.endswith("1.bias"): t.copy_(0.1 * torch.randn_like(t))
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in head.state_dict().items()}
    feats = (torch.randn(N, 7, 11), torch.randn(N, in_ch, Lf))
    mask = (torch.rand(N, ch, Lf) >= p)
    head = head.to(dev)
    head.fixed_dropout_mask = mask.to(dev, torch.uint8)
    x = feats[1].to(dev).requires_grad_(True)
    xr = feats[1].clone().requires_grad_(True)
    head.eval()
    with torch.no_grad():
        ye = head((None, x))
    assert rel(ye, _ref_head(sd, (None, xr), -1, num_convs, concat_input, 1, 1, False, None, p)) < 2e-5
    head.train()
    y = head((None, x))
    yr = _ref_head(sd, (None, xr), -1, num_convs, concat_input, 1, 1, True, mask.float(), p)
    assert rel(y, yr) < 2e-5
    dy = torch.randn_like(yr)
    y.backward(dy.to(dev)); yr.backward(dy)
    assert rel(x.grad, xr.grad) < 5e-5
    for name, t in head.named_parameters():
        assert rel(t.grad, sd[name].grad) < 5e-5, name


@pytest.mark.parametrize("num_convs,concat_input", [(1, True), (2, True), (0, False), (2, False)])
def test_fcn_head_variants_under_use_amp(num_convs, concat_input, dev):
    """``use_amp: true`` (the reference's default) with a head other than the shipped (num_convs 1, concat_input false): the plugin
    must train, not raise (round-5 verdict).  ``concat_input`` / ``num_convs: 0`` take the backbone's blocked bf16 feature, convert it to
    fp32 (exact) and run the fp32 general form - more precise than autocast's 16-bit head; ``num_convs: 2`` without concat chains two
    bf16 conv units.  One step of ``base.train_one_epoch(use_amp=True)``: the head's output against a torch-CPU restatement of
    fcn_head.py:89-97 on the feature the backbone handed over (fp32 bar for the fp32 forms, the 16-bit bar for the bf16 units), the
    loss within 2 % of the fp32 run's, every parameter with a finite gradient and moved by the optimiser."""
    import copy
    import algorithms.base as A_base
    from ssecg import amp as SAMP
    from utils.misc import NativeScalerWithGradNormCount
    from utils.optimizer import get_optimizer_from_config
    C, B, L, p = 2, 4, 2000, 0.1
    cfg = model_cfg(C)
    hk = cfg["decode_head"]["FCNHead"]
    hk.update(num_convs=num_convs, concat_input=concat_input)
    if num_convs == 0:
        hk["channels"] = hk["in_channels"]
    torch.manual_seed(11)
    model = A_base.init_model_from_cfg(cfg).to(dev)
    twin = copy.deepcopy(model)
    ch = hk["channels"]
    mask = torch.from_numpy(synth.uniform(3, 77, B * ch * 63).reshape(B, ch, 63) >= p)
    model.decode_head.fixed_dropout_mask = mask.to(dev, torch.uint8)
    twin.decode_head.fixed_dropout_mask = mask.to(dev, torch.uint8)
    sd = {k: v.detach().cpu().clone() for k, v in model.decode_head.state_dict().items()}
    batch = {k: torch.from_numpy(v).to(dev) for k, v in synth.learnable_batch(77, B, C, L)["labeled"].items()}
    cap = {}

    def hook(mod, inp, out):
        f = inp[0][mod.in_index]
        cap["blocked"] = SAMP.is_blocked(f)
        cap["feat"] = (SAMP.to_planar(f.detach()) if cap["blocked"] else f.detach()).cpu()
        cap["out"] = out.detach().cpu()

    h = model.decode_head.register_forward_hook(hook)
    before = {k: v.detach().clone() for k, v in model.named_parameters()}
    grads = {}
    for k, prm in model.named_parameters():
        prm.register_hook(lambda gr, k=k: grads.__setitem__(k, gr.detach().clone()))
    tcfg = dict(TRAIN_CFG)
    st = A_base.train_one_epoch(model, [batch], get_optimizer_from_config(tcfg, model.parameters()), dev, 3, NativeScalerWithGradNormCount(),
                                None, True, tcfg)
    h.remove()
    assert cap["blocked"], "the backbone did not hand the head a blocked bf16 feature under use_amp"
    ref = _ref_head(sd, (cap["feat"],), -1, num_convs, concat_input, 1, 1, True, mask.float(), p)
    lp_units = num_convs >= 1 and not concat_input
    assert rel(cap["out"], ref) < (2e-2 if lp_units else 2e-5), rel(cap["out"], ref)
    st32 = A_base.train_one_epoch(twin, [batch], get_optimizer_from_config(tcfg, twin.parameters()), dev, 3, NativeScalerWithGradNormCount(),
                                  None, False, tcfg)
    assert np.isfinite(st["loss"]) and abs(st["loss"] - st32["loss"]) < 2e-2 * st32["loss"], (st["loss"], st32["loss"])
    assert set(grads) == set(before)
    for k, prm in model.named_parameters():
        assert torch.isfinite(grads[k]).all(), k
        assert not torch.equal(prm.detach(), before[k]), k


def test_evaluate_fast_path_and_test_artifacts(dev, tmp_path):
    """evaluate(return_outputs=False) gives the same statistics as the full form without the probabilities; test()
    writes the reference's artefacts: test_metrics.csv (one row, %.4f, metric columns + loss), test_outputs.npy,
    test_labels.npy (src/algorithms/base.py:484-498)."""
    import algorithms.base as A_base
    import utils.misc as misc
    C, B = 2, 4
    model = build_hip_model(C, synth.model_state(9, C, trained=True, sharpen=16.0), dev)
    loader = [{"ecg": torch.from_numpy(b["ecg"]), "target": torch.from_numpy(b["target"])}
              for b in (synth.fixmatch_batch(40 + i, B, C, 2000)["labeled"] for i in range(3))]
    s1, m1, o1, l1 = A_base.evaluate(model, loader, dev, None, use_amp=False)
    s2, m2, o2, l2 = A_base.evaluate(model, loader, dev, None, use_amp=False, return_outputs=False)
    assert o2 is None and l2 is None and s1 == s2 and m1 == m2
    assert tuple(o1.shape) == (3 * B, 4, 2000) and tuple(l1.shape) == (3 * B, 4, 2000)
    assert torch.allclose(o1.sum(dim=1), torch.ones(3 * B, 2000), atol=1e-5) and l1.sum(dim=1).eq(1).all()
    # loss meter = mean over records of the per-batch CE
    ref = np.mean([F.cross_entropy(torch.log(o1[i * B:(i + 1) * B].double()), torch.from_numpy(
        synth.fixmatch_batch(40 + i, B, C, 2000)["labeled"]["target"])).item() for i in range(3)])
    assert abs(s1["loss"] - ref) < 1e-4 * ref
    # test(): checkpoint -> artefacts
    out_dir = str(tmp_path)
    cfg = dict(model_cfg(C))
    cfg.update({"output_dir": out_dir, "exp_name": "exp", "device": "cuda:0", "use_amp": False, "algorithm": "base",
                "dataset": {"synthetic": {"num_leads": C, "num_test": 6, "seed": 5}, "signal_length": 2000},
                "dataloader": {"batch_size": 4, "num_workers": 0}, "ddp": {}, "test": {"target_metric": "MeanIoU"},
                "metric": {"task": "segmentation", "num_classes": 4, "target_metrics": ["MeanIoU"]}})
    os.makedirs(os.path.join(out_dir, "exp"), exist_ok=True)
    misc.save_model(cfg, os.path.join(out_dir, "exp", "best-MeanIoU.pth"), 0, model)
    metrics = A_base.test(cfg)
    d = os.path.join(out_dir, "exp")
    import pandas as pd
    df = pd.read_csv(os.path.join(d, "test_metrics.csv"))
    assert list(df.columns) == ["MeanIoU", "loss"] and len(df) == 1
    assert abs(df["MeanIoU"][0] - metrics["MeanIoU"]) < 1e-4 and abs(df["loss"][0] - metrics["loss"]) < 1e-4
    outs, labs = np.load(os.path.join(d, "test_outputs.npy")), np.load(os.path.join(d, "test_labels.npy"))
    assert outs.shape == (6, 4, 2000) and labs.shape == (6, 4, 2000) and outs.dtype == np.float32
    # --model_path override (src/test.py:62-66)
    cfg["test"] = {"model_path": os.path.join(d, "best-MeanIoU.pth")}
    assert abs(A_base.test(cfg)["MeanIoU"] - metrics["MeanIoU"]) < 1e-12
    # inference.py (src/inference.py:76-126): same checkpoint, probabilities only, auxiliary-head entries dropped
    import inference as INF
    ck = torch.load(cfg["test"]["model_path"], map_location="cpu", weights_only=False)
    ck["model"]["auxiliary_head.0.weight"] = torch.zeros(3)
    torch.save(ck, os.path.join(d, "with_aux.pth"))
    cfg["test"] = {"model_path": os.path.join(d, "with_aux.pth")}
    probs = INF.inference(cfg)
    assert probs.shape == (6, 4, 2000) and probs.dtype == np.float32
    assert np.abs(probs - outs).max() < 1e-6 and np.array_equal(np.load(os.path.join(d, "test_outputs.npy")), probs)


# ----------------------------------------------------------------------------- standalone BatchNorm1d / ReLU modules
@pytest.mark.parametrize("shape", [(6, 64, 125), (3, 5, 37)])
@pytest.mark.parametrize("training", [True, False], ids=["train", "eval"])
def test_standalone_batchnorm_and_relu_modules(shape, training, dev):
    """``ssecg.nn.BatchNorm1d`` / ``ReLU`` called on their own (a hook, a custom head) behave like torch.nn's: output,
    running statistics, num_batches_tracked and all gradients (round 1 raised NotImplementedError here)."""
    from ssecg import nn as SN
    torch.manual_seed(3)
    N, C, L = shape
    x = torch.randn(N, C, L) * 1.5 + 0.3
    ref_bn, ref_relu = torch.nn.BatchNorm1d(C), torch.nn.ReLU()
    with torch.no_grad():
        ref_bn.weight.uniform_(0.5, 1.5); ref_bn.bias.normal_(0, 0.2)
        ref_bn.running_mean.normal_(0, 0.3); ref_bn.running_var.uniform_(0.5, 2.0)
    bn, relu = SN.BatchNorm1d(C).to(dev), SN.ReLU()
    bn.load_state_dict(ref_bn.state_dict())
    ref_bn.train(training); bn.train(training)
    xr = x.clone().double().requires_grad_(True)
    ref_bn = ref_bn.double()
    yr = ref_relu(ref_bn(xr))
    g = torch.randn(N, C, L)
    yr.backward(g.double())
    xd = x.to(dev).requires_grad_(True)
    y = relu(bn(xd))
    y.backward(g.to(dev))
    sc = yr.abs().max().item()
    assert (y.detach().cpu().double() - yr.detach()).abs().max().item() < 2e-5 * sc
    assert (xd.grad.cpu().double() - xr.grad).abs().max().item() < 5e-5 * xr.grad.abs().max().item()
    assert (bn.weight.grad.cpu().double() - ref_bn.weight.grad).abs().max().item() < 5e-5 * ref_bn.weight.grad.abs().max().item()
    assert (bn.bias.grad.cpu().double() - ref_bn.bias.grad).abs().max().item() < 5e-5 * ref_bn.bias.grad.abs().max().item()
    assert torch.allclose(bn.running_mean.cpu().double(), ref_bn.running_mean, rtol=1e-5, atol=1e-6)
    assert torch.allclose(bn.running_var.cpu().double(), ref_bn.running_var, rtol=1e-5, atol=1e-6)
    assert int(bn.num_batches_tracked) == int(ref_bn.num_batches_tracked)


def test_device_prefetcher_feeds_the_same_batches(dev):
    """Pinned HOST batches through utils/semi_dataset.DevicePrefetcher (next batch copied on a side stream under the current
    step): every yielded tensor is on the device and equal to its host original, nested structures and non-tensor entries are
    preserved, and a FixMatch epoch fed through it is bit-identical to one fed device-resident batches."""
    import algorithms.fixmatch as A_fm
    from utils.misc import NativeScalerWithGradNormCount
    from utils.optimizer import get_optimizer_from_config
    from utils.semi_dataset import DevicePrefetcher
    C, B, L = 2, 2, 2000
    host = []
    for i in range(4):
        b = synth.fixmatch_batch(300 + i, B, C, L)
        host.append(({k: torch.from_numpy(v).pin_memory() for k, v in b["labeled"].items()},
                     {**{k: torch.from_numpy(v).pin_memory() for k, v in b["unlabeled"].items()}, "ids": [i, i + 1]}))
    seen = list(DevicePrefetcher(host, dev))
    assert len(seen) == 4
    for (hl, hu), (dl, du) in zip(host, seen):
        assert du["ids"] == hu["ids"]
        for hb, db in ((hl, dl), (hu, du)):
            for k, v in hb.items():
                if torch.is_tensor(v):
                    assert db[k].device == dev and torch.equal(db[k].cpu(), v), k
    res = {}
    for mode in ("resident", "prefetched"):
        model = build_hip_model(C, synth.model_state(77, C, trained=True, sharpen=1.0), dev)
        model.decode_head.dropout = None; model.decode_head.dropout_ratio = 0.0
        opt = get_optimizer_from_config(dict(TRAIN_CFG), model.parameters())
        lab, unl = [h[0] for h in host], [{k: v for k, v in h[1].items() if k != "ids"} for h in host]
        if mode == "resident":
            lab = [{k: v.to(dev) for k, v in b.items()} for b in lab]; unl = [{k: v.to(dev) for k, v in b.items()} for b in unl]
        stats = A_fm.train_one_epoch(model, lab, unl, opt, dev, 3, NativeScalerWithGradNormCount(), None, False,
                                     dict(TRAIN_CFG, conf_thresh=0.3))
        res[mode] = (stats, {k: v.detach().clone() for k, v in model.state_dict().items()})
    assert res["resident"][0] == res["prefetched"][0]
    for k, v in res["resident"][1].items():
        assert torch.equal(v, res["prefetched"][1][k]), k

