"""GPU: checkpoint / resume (src/utils/misc.py:281-321) - a run interrupted after a step, saved with the reference's
checkpoint schema and resumed into fresh objects continues bit-identically (weights, BN buffers, EMA teacher, AdamW
moments), for the FixMatch and MeanTeacher plugins."""
import os

import numpy as np
import pytest
import torch

from helpers import L, TRAIN_CFG, build_hip_model, dropout_mask_np, model_cfg, sharpen_for, to_dev
from ssecg import synth

pytestmark = pytest.mark.gpu


def _step(algo, model, teacher, optimizer, scaler, dev, s, seed, C, B, cfg):
    import algorithms.fixmatch as A_fm
    import algorithms.mean_teacher as A_mt
    batch = to_dev(synth.fixmatch_batch(seed + 10 + s, B, C, L), dev)
    model.decode_head.fixed_dropout_mask = torch.from_numpy(dropout_mask_np(seed + 10 + s, 2 * B)).to(dev, torch.uint8)
    if algo == "fixmatch":
        return A_fm.train_one_epoch(model, [batch["labeled"]], [batch["unlabeled"]], optimizer, dev, 3 + 9 * s, scaler, None,
                                    False, cfg)
    return A_mt.train_one_epoch(model, teacher, [batch["labeled"]], [batch["unlabeled"]], optimizer, dev, 3 + 9 * s, scaler,
                                None, False, cfg)


def _fresh(algo, C, seed, dev, cfg, init_seed=None):
    import algorithms.mean_teacher as A_mt
    import utils.misc as misc
    from utils.optimizer import get_optimizer_from_config
    model = build_hip_model(C, synth.model_state(seed if init_seed is None else init_seed, C, trained=True,
                                                 sharpen=sharpen_for(C)), dev)
    teacher = A_mt.make_teacher(model_cfg(C), model, dev) if algo == "mean_teacher" else None
    optimizer = get_optimizer_from_config(cfg, model.parameters())
    return model, teacher, optimizer, misc.NativeScalerWithGradNormCount()


def _assert_optimizer_equal(oA, oC):
    stA, stC = oA.state_dict(), oC.state_dict()
    assert stA["param_groups"] == stC["param_groups"]
    for i in stA["state"]:
        for k, v in stA["state"][i].items():
            assert torch.equal(torch.as_tensor(v).cpu(), torch.as_tensor(stC["state"][i][k]).cpu()), (i, k)


def test_fixmatch_resume_continues_bit_identically(dev, tmp_path):
    import utils.misc as misc
    algo, C, B, seed = "fixmatch", 1, 2, 21
    cfg = dict(TRAIN_CFG)
    mA, _, oA, sA = _fresh(algo, C, seed, dev, cfg)                     # uninterrupted: two steps
    for s in range(2):
        statsA = _step(algo, mA, None, oA, sA, dev, s, seed, C, B, cfg)
    mB, _, oB, sB = _fresh(algo, C, seed, dev, cfg)                     # interrupted after step 0
    _step(algo, mB, None, oB, sB, dev, 0, seed, C, B, cfg)
    path = os.path.join(tmp_path, "ck.pth")
    misc.save_model({"resume": None, "note": "x"}, path, 0, mB, oB, sB, metrics={"loss": 1.0})
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert set(ck) == {"epoch", "model", "optimizer", "scaler", "config", "metrics"}
    mC, _, oC, sC = _fresh(algo, C, seed, dev, cfg, init_seed=seed + 99)  # resumed into differently initialised objects
    run_cfg = {"resume": path, "start_epoch": 0}
    misc.load_model(run_cfg, mC, oC, sC)
    assert run_cfg["start_epoch"] == 1
    statsC = _step(algo, mC, None, oC, sC, dev, 1, seed, C, B, cfg)
    for k in statsA:
        assert statsA[k] == statsC[k], k
    sdA, sdC = mA.state_dict(), mC.state_dict()
    for k in sdA:
        assert torch.equal(sdA[k], sdC[k]), k
    _assert_optimizer_equal(oA, oC)


def test_mean_teacher_resume_follows_the_reference(dev, tmp_path):
    """The reference builds the teacher with its parameters bound to the student's storage (mean_teacher.py:281-290)
    BEFORE load_model; load_model then loads ``model`` and afterwards ``model_ema`` (misc.py:313-315) - through the
    shared storage the student therefore restarts from the EMA weights.  Same here, by construction."""
    import utils.misc as misc
    algo, C, B, seed = "mean_teacher", 2, 2, 23
    cfg = dict(TRAIN_CFG)
    mB, tB, oB, sB = _fresh(algo, C, seed, dev, cfg)
    for s in range(2):
        _step(algo, mB, tB, oB, sB, dev, s, seed, C, B, cfg)
    path = os.path.join(tmp_path, "ck.pth")
    misc.save_model({"resume": None}, path, 7, mB, oB, sB, metrics={"loss": 1.0}, model_ema=tB)
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert set(ck) == {"epoch", "model", "optimizer", "scaler", "config", "metrics", "model_ema"}
    assert ck["model_ema"]["backbone.stem.1.num_batches_tracked"].dtype == torch.float32          # Q5
    k0 = "backbone.layer1.0.conv1.weight"
    assert not torch.equal(ck["model"][k0], ck["model_ema"][k0])                                  # a real EMA by now
    mC, tC, oC, sC = _fresh(algo, C, seed, dev, cfg, init_seed=seed + 99)
    run_cfg = {"resume": path, "start_epoch": 0}
    misc.load_model(run_cfg, mC, oC, sC, model_ema=tC)
    assert run_cfg["start_epoch"] == 8
    names = [k for k, _ in mC.named_parameters()]
    for (k, ps), pt in zip(mC.named_parameters(), tC.parameters()):
        assert ps.data_ptr() == pt.data_ptr(), k                                                   # still bound
        assert torch.equal(ps.detach().cpu(), ck["model_ema"][k]), k                              # = EMA weights
    tsd = tC.state_dict()
    for k, v in ck["model_ema"].items():
        if k not in names:
            assert tsd[k].dtype == v.dtype and torch.equal(tsd[k].cpu(), v), k
    msd = mC.state_dict()
    for k, v in ck["model"].items():
        if k not in names:
            assert torch.equal(msd[k].cpu(), v), k                                                # student buffers: its own
    _assert_optimizer_equal(oB, oC)
    stats = _step(algo, mC, tC, oC, sC, dev, 2, seed, C, B, cfg)
    assert all(np.isfinite(v) for v in stats.values())
    assert next(mC.parameters()).data_ptr() != next(tC.parameters()).data_ptr()                  # un-bound by the first EMA


def _train_cfg(out_dir, hip_graph, epochs, resume=None, use_amp=False):
    C = 2
    cfg = dict(model_cfg(C))
    cfg.update({"seed": 0, "output_dir": out_dir, "exp_name": "run", "resume": resume, "start_epoch": 0, "device": "cuda:0",
                "use_amp": use_amp, "algorithm": "fixmatch", "mode": "scratch", "pretrained_backbone": None,
                # 72 unlabelled windows in batches of 16 with drop_last false: four full batches and a SHORT last one (8) per epoch
                "dataset": {"synthetic": {"num_leads": C, "num_unlabeled": 72, "num_labeled": 16, "num_valid": 24, "num_test": 8, "seed": 3},
                            "signal_length": 1000},
                "dataloader": {"batch_size": 16, "num_workers": 0, "drop_last": False},
                "train": dict(TRAIN_CFG, epochs=epochs, warmup_epochs=1, blr=None, conf_thresh=0.3),
                "metric": {"task": "segmentation", "num_classes": 4, "target_metrics": ["MeanIoU"]},
                "ddp": {"world_size": 1, "rank": -1, "gpu": 0, "dist_url": "env://", "dist_backend": "nccl", "distributed": False,
                        "sync_bn": True}})
    if hip_graph is not None:
        cfg["train"]["hip_graph"] = hip_graph
    return cfg


@pytest.mark.parametrize("use_amp", [False, True], ids=["fp32", "bf16"])
def test_train_loop_under_the_auto_graph_default_equals_the_eager_loop(use_amp, dev, tmp_path, monkeypatch):
    """``train.hip_graph`` defaults to ``auto`` = ON for the reference's shipped batch size (16 windows per loader), so every
    shipped config trains through capture + replay with the two-stream fork / join inside the capture.  The plugin's real ``train(config)``
    - FixMatch, batch 16, ``drop_last: false`` (a short last batch every epoch: eager fall-back between replays), ``evaluate()``
    between the epochs (on the 16-bit eval path under use_amp), a checkpoint after epoch 0 and a RESUMED epoch 1 - under the auto
    default against ``hip_graph: false``: the same logged statistics, validation loss and final weights / AdamW moments, bit for bit."""
    import algorithms.base as A_base
    import algorithms.fixmatch as A_fm
    import utils.misc as misc
    from ssecg.graph import StepGraph
    runs = {}
    for mode in ("auto", False):
        out = os.path.join(tmp_path, f"g_{mode}")
        seen = {"tails": [], "replays": 0}
        orig_tail = A_base.epoch_tail

        def tail(config, output_dir, log_writer, epoch, model_without_ddp, optimizer, loss_scaler, train_stats, valid_stats, metrics, best,
                 **kw):
            seen["tails"].append((epoch, dict(train_stats), dict(valid_stats), dict(metrics)))
            seen["state"] = {k: v.detach().clone() for k, v in model_without_ddp.state_dict().items()}
            seen["moments"] = [v["exp_avg"].detach().clone() for v in optimizer.state_dict()["state"].values()]
            g = getattr(model_without_ddp, "_ssecg_step_graph", None)
            seen["replays"] = max(seen["replays"], g.replays if g is not None else 0)
            misc.save_model(config, os.path.join(output_dir, "last.pth"), epoch, model_without_ddp, optimizer, loss_scaler, metrics={})
            return orig_tail(config, output_dir, log_writer, epoch, model_without_ddp, optimizer, loss_scaler, train_stats, valid_stats,
                             metrics, best, **kw)

        monkeypatch.setattr(A_fm, "epoch_tail", tail)
        torch.manual_seed(0)
        cfg = _train_cfg(out, None if mode == "auto" else False, 1, use_amp=use_amp)      # epoch 0, then stop
        A_fm.train(cfg)
        assert cfg["train"]["hip_graph"] is (mode == "auto")
        cfg2 = _train_cfg(out, None if mode == "auto" else False, 2, resume=os.path.join(out, "run", "last.pth"), use_amp=use_amp)
        A_fm.train(cfg2)                                                                   # resumed: epoch 1
        assert cfg2["start_epoch"] == 1 and [t[0] for t in seen["tails"]] == [0, 1]
        if mode == "auto":
            assert seen["replays"] >= 2, "the auto default never replayed a graph"
        runs[mode] = seen
    a, e = runs["auto"], runs[False]
    for (ea, ta, va, ma), (ee, te, ve, me) in zip(a["tails"], e["tails"]):
        assert ta == te and va == ve and ma == me, (ea, ta, te, va, ve)
    for k, v in a["state"].items():
        assert torch.equal(v, e["state"][k]), k
    for x, y in zip(a["moments"], e["moments"]):
        assert torch.equal(x, y)
