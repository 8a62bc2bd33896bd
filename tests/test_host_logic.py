"""CPU: host-side logic of the product package (no kernels): schedules, meters, metric buffer, config merge,
synthetic data determinism, checkpoint schema, registries."""
import io
import math
import os

import numpy as np
import pytest
import torch

from oracle import torch_ref as O
from ssecg import synth


def test_lr_schedule_matches_oracle():
    import utils.lr_sched as S
    cfg = dict(warmup_epochs=10, epochs=100, lr=1e-3, min_lr=1e-4)
    for e in (0.0, 0.5, 3.0, 9.99, 10.0, 12.0, 55.5, 99.9):
        assert S.lr_at(e, cfg) == O.lr_at(e, cfg)
    class Opt: param_groups = [{"lr": 0.0}, {"lr": 0.0, "lr_scale": 0.5}]
    lr = S.adjust_learning_rate(Opt, 12.0, cfg)
    assert Opt.param_groups[0]["lr"] == lr and Opt.param_groups[1]["lr"] == lr * 0.5


def test_synth_is_deterministic_and_shaped():
    a, b = synth.fixmatch_batch(7, 3, 2, 2000), synth.fixmatch_batch(7, 3, 2, 2000)
    for g in a:
        for k in a[g]:
            assert np.array_equal(a[g][k], b[g][k])
    assert a["labeled"]["ecg"].shape == (3, 2, 2000) and a["labeled"]["target"].dtype == np.int64
    t = a["labeled"]["target"]
    assert t.min() >= 0 and t.max() <= 3
    runs = np.diff(np.flatnonzero(np.diff(t[0]) != 0))
    assert runs.min() >= 50 if len(runs) else True
    x = synth.normal(1, 1, (200000,))
    assert abs(x.mean()) < 0.01 and abs(x.std() - 1) < 0.01
    sd = synth.model_state(0, 12)
    assert sd["backbone.stem.0.weight"].shape == (64, 12, 7) and len(sd) == 128
    assert len(synth.param_keys(sd)) == 65 and len(synth.buffer_keys(sd)) == 63


def test_state_dict_keys_match_reference_layout():
    from helpers import build_hip_model
    sd_np = synth.model_state(3, 2)
    model = build_hip_model(2, sd_np, torch.device("cpu"))
    assert list(model.state_dict().keys()) == list(sd_np.keys())
    assert sum(p.numel() for p in model.parameters()) == 4_041_284 + 448
    for k, v in model.state_dict().items():
        assert tuple(v.shape) == sd_np[k].shape, k


def test_metric_logger_and_device_buffer():
    import utils.misc as misc
    ml = misc.MetricLogger(delimiter="  ")
    buf = misc.DeviceMetricBuffer(["a", "b"], 2, torch.device("cpu"))
    for i in range(5):  # grows past its initial capacity
        buf.push(torch.tensor([float(i), 2.0 * i]))
    rows = buf.flush(ml)
    assert len(rows) == 5 and rows[3] == {"a": 3.0, "b": 6.0}
    assert ml.meters["a"].global_avg == 2.0 and ml.meters["b"].median == 4.0
    assert buf.flush(ml) == []
    buf.push(torch.tensor([float("nan"), 0.0]))
    with pytest.raises(SystemExit):
        buf.flush(ml)
    out = [x for x in ml.log_every(range(3), 2, "hdr")]
    assert out == [0, 1, 2]


def test_config_merge_and_registries(tmp_path):
    import train as T
    base = {"a": 1, "train": {"lr": 1.0, "epochs": 3}, "backbone": {"resnet18": {"num_leads": 1}}}
    T.deep_merge(base, {"train": {"lr": 2.0}, "exp_name": "x"})
    assert base == {"a": 1, "train": {"lr": 2.0, "epochs": 3}, "backbone": {"resnet18": {"num_leads": 1}}, "exp_name": "x"}
    import algorithms
    for name in ("base", "fixmatch", "mean_teacher"):
        mod = algorithms.__dict__[name]
        assert callable(mod.train) and callable(mod.test) and callable(mod.train_one_epoch)
    import models.backbones as bb
    import models.decode_heads as dh
    assert "resnet18" in bb.__dict__ and "FCNHead" in dh.__dict__
    with pytest.raises(NotImplementedError):
        bb.resnet50(num_leads=1)
    with pytest.raises(ValueError):
        T.main({"algorithm": "reco"})


def test_checkpoint_schema_roundtrip(tmp_path):
    import utils.misc as misc
    from helpers import build_hip_model
    model = build_hip_model(1, synth.model_state(0, 1), torch.device("cpu"))
    scaler = misc.NativeScalerWithGradNormCount()
    path = os.path.join(tmp_path, "ck.pth")
    misc.save_model({"resume": None}, path, 4, model, None, scaler, metrics={"loss": 1.0}, model_ema=model)
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert set(ck) == {"epoch", "model", "optimizer", "scaler", "config", "metrics", "model_ema"}
    assert set(ck["scaler"]) >= {"scale", "growth_factor", "backoff_factor", "growth_interval", "_growth_tracker"}
    m2 = build_hip_model(1, synth.model_state(9, 1), torch.device("cpu"))
    cfg = {"resume": path, "eval": True}
    misc.load_model(cfg, m2, None, scaler)
    assert torch.equal(m2.state_dict()["backbone.stem.0.weight"], model.state_dict()["backbone.stem.0.weight"])


def test_synthetic_loaders():
    from utils.semi_dataset import build_seg_dataset, get_dataloader
    cfg = {"synthetic": {"num_unlabeled": 10, "num_labeled": 4, "num_leads": 2}, "signal_length": 2000}
    du = build_seg_dataset(cfg, "train_unlabeled")
    dl = build_seg_dataset(cfg, "train_labeled", num_unlabeled=len(du))
    assert len(du) == len(dl) == 10
    b = next(iter(get_dataloader(du, mode="train", batch_size=4)))
    assert b["ecg"].shape == (4, 2, 2000) and b["ecg_aug"].shape == (4, 2, 2000)
    b = next(iter(get_dataloader(dl, mode="train", batch_size=4)))
    assert b["target"].shape == (4, 2000) and b["target"].dtype == torch.int64
    with pytest.raises(NotImplementedError):
        build_seg_dataset({"ecg_dir": "/x"}, "valid")


def test_fused_adamw_state_dict_layout():
    from ssecg.optim import FusedAdamW
    p = torch.nn.Parameter(torch.zeros(3))
    opt = FusedAdamW([p], lr=1e-3, betas=(0.9, 0.999), weight_decay=0.05)
    ref = torch.optim.AdamW([torch.nn.Parameter(torch.zeros(3))], lr=1e-3, weight_decay=0.05)
    assert set(opt.state_dict()["param_groups"][0]) >= {"lr", "betas", "eps", "weight_decay", "params"}
    opt.load_state_dict(ref.state_dict())  # a reference AdamW checkpoint loads
    p.grad = torch.ones(3)
    from ssecg.lib import SsecgError
    with pytest.raises(SsecgError):
        opt.step()  # CPU parameter: fails loudly


def test_device_prefetcher_is_transparent_off_device():
    """utils/semi_dataset.DevicePrefetcher on a CPU 'device': same batches, same order, same length, wrapped loader reachable."""
    from utils.semi_dataset import DevicePrefetcher, device_prefetch
    batches = [{"ecg": torch.full((2, 1, 8), float(i)), "target": torch.zeros(2, 8, dtype=torch.long), "id": f"r{i}"} for i in range(3)]
    assert device_prefetch(batches, "cpu") is batches and device_prefetch(batches, "cuda", enabled=False) is batches
    pf = DevicePrefetcher(batches, "cpu")
    assert len(pf) == 3 and [b["id"] for b in pf] == ["r0", "r1", "r2"]
    assert all(torch.equal(a["ecg"], b["ecg"]) for a, b in zip(pf, batches))
    assert pf.count({"x": 1}) == 0          # attribute access falls through to the wrapped loader (here: list.count)


REF_SRC = "/root/reference/src"


@pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="the reference checkout exists in the build container only")
def test_reference_record_pipeline_through_the_dataset_hook(tmp_path):
    """``dataset.reference_src`` (utils/semi_dataset.py:load_reference_pipeline): the reference's OWN record pipeline
    (src/utils/semi_dataset.py:29-322, src/utils/transforms.py), loaded where it lies under private module names, fed a small
    record set in its on-disk format - pickled single-lead waveforms + labels and index CSVs (semi_dataset.py:209-244,260-277) -
    through the product's ``build_seg_dataset`` / ``get_dataloader``: the batch contract the plugins rely on."""
    import pandas as pd
    import pickle
    import sys
    from utils.semi_dataset import build_seg_dataset, get_dataloader
    rng = np.random.default_rng(0)
    ecg_dir, lab_dir, idx_dir = tmp_path / "ecg", tmp_path / "lab", tmp_path / "idx"
    for d in (ecg_dir, lab_dir, idx_dir):
        d.mkdir()
    T = 2500
    names = [f"rec{i:03d}.pkl" for i in range(12)]
    for nme in names:
        with open(ecg_dir / nme, "wb") as f:
            pickle.dump(rng.standard_normal(T), f)
        with open(lab_dir / nme, "wb") as f:
            pickle.dump(np.repeat(rng.integers(0, 4, T // 100), 100), f)
    pd.DataFrame({"waveform": names[:4], "label": names[:4]}).to_csv(idx_dir / "lab.csv", index=False)
    pd.DataFrame({"waveform": names[4:]}).to_csv(idx_dir / "unl.csv", index=False)
    pd.DataFrame({"waveform": names[:4], "label": names[:4]}).to_csv(idx_dir / "val.csv", index=False)
    # the shipped configuration's own pipeline (configs/base/resnet18/fixmatch.yaml:46-81): resample to signal_length, Butterworth
    # band-pass, FFT resize-crop (weak), RandAugment of four noise ops (strong), standardise, to tensor
    cfg = {"reference_src": REF_SRC, "index_dir": str(idx_dir), "ecg_dir": str(ecg_dir), "label_dir": str(lab_dir),
           "filename_col": "waveform", "label_filename_col": "label", "train_labeled_csv": "lab.csv", "train_unlabeled_csv": "unl.csv",
           "valid_csv": "val.csv", "signal_length": 2000,
           "filter": [{"highpass_filter": {"fs": 250, "cutoff": 0.67}}, {"lowpass_filter": {"fs": 250, "cutoff": 40}}],
           "augmentations": [{"random_resize_crop": {"target_length": 2000, "scale_min": 0.5, "scale_max": 2.0}}],
           "strong_augmentations": [{"RandAugment": {"ops": [{"AmplitudeScaling": {"sigma": 0.5}}, {"AdaptivePowerlineNoise": {"fs": 250}},
                                                             {"RandomPartialWhiteNoise": {"amplitude": 1, "ratio": 0.5}},
                                                             {"RandomPartialSineNoise": {"amplitude": 1, "ratio": 0.5}}],
                                                     "level": 10, "num_layers": 3, "prob": 0.5}}],
           "transforms": [{"standardize": {"axis": [-1, -2]}}, {"to_tensor": {"dtype": "float"}}]}
    import utils as own_utils
    ds_u = build_seg_dataset(cfg, split="train_unlabeled")
    ds_l = build_seg_dataset(cfg, split="train_labeled", num_unlabeled=len(ds_u))
    ds_v = build_seg_dataset(cfg, split="valid")
    assert type(ds_u).__module__ == "_ssecg_reference_semi_dataset"             # the reference's class, not a re-implementation
    assert sys.modules["utils"] is own_utils and "transforms" not in vars(own_utils)   # the product's package is untouched
    assert len(ds_u) == 8 and len(ds_l) == 8 and len(ds_v) == 4                 # labelled set over-sampled to the unlabelled length (:86-95)
    bu = next(iter(get_dataloader(ds_u, mode="train", batch_size=4, num_workers=0)))
    bl = next(iter(get_dataloader(ds_l, mode="train", batch_size=4, num_workers=0)))
    bv = next(iter(get_dataloader(ds_v, mode="valid", batch_size=2, num_workers=0)))
    assert set(bu) == {"ecg", "ecg_aug"} and bu["ecg"].shape == (4, 1, 2000) and bu["ecg"].dtype == torch.float32
    assert bu["ecg_aug"].shape == (4, 1, 2000) and not torch.equal(bu["ecg"], bu["ecg_aug"])
    assert {"ecg", "target"} <= set(bl) and bl["target"].shape == (4, 2000) and bl["target"].dtype == torch.int64   # (the labelled split gets the strong view too)
    assert bv["ecg"].shape == (2, 1, 2000) and int(bl["target"].max()) <= 3
    assert abs(float(bu["ecg"].mean())) < 1e-4 and abs(float(bu["ecg"].std()) - 1.0) < 1e-2      # Standardize (transforms.py:290-310)


def test_hip_graph_auto_resolution():
    """``train.hip_graph`` absent / ``auto``: on for the runs the eager loop cannot feed (<= 128 windows per loader, one GPU, no
    accumulation - the reference's shipped batch_size 16), off otherwise; explicit values are taken as given."""
    from algorithms.base import resolve_hip_graph

    def cfg(bs, dist=False, accum=1, hip=None, device="cuda", backend="nccl", reducer=None):
        c = {"device": device, "dataloader": {"batch_size": bs}, "ddp": {"distributed": dist, "dist_backend": backend},
             "train": {"accum_iter": accum}}
        if reducer:
            c["ddp"]["reducer"] = reducer
        if hip is not None:
            c["train"]["hip_graph"] = hip
        resolve_hip_graph(c)
        return c["train"]["hip_graph"]

    assert cfg(16) is True and cfg(128) is True and cfg(129) is False and cfg(256) is False
    assert cfg(16, accum=2) is False and cfg(16, device="cpu") is False
    # distributed (round 6): on over RCCL with this library's reducer - the collectives are captured with the step; gloo / torch's
    # DistributedDataParallel keep the eager loop
    assert cfg(16, dist=True) is True and cfg(256, dist=True) is False
    assert cfg(16, dist=True, backend="gloo") is False and cfg(16, dist=True, reducer="torch") is False
    assert cfg(16, hip=False) is False and cfg(512, hip=True) is True and cfg(16, hip="auto") is True


def test_pass_overlap_is_off_for_wrappers_that_broadcast_buffers(monkeypatch):
    """The side-stream pseudo-label pass must not run when a data-parallel wrapper broadcasts its buffers at every forward
    (``ddp.sync_bn: false``): that collective - and its write into the running statistics - would be issued from the side stream.
    The group size comes from the wrapper's process group, not from a ``world_size`` attribute torch's DistributedDataParallel does
    not have (the real wrappers at two ranks: tests/test_dist_gloo.py (4b))."""
    import types
    import torch.distributed as dist
    from ssecg import ops
    dev = types.SimpleNamespace(type="cuda", index=0)
    plain = types.SimpleNamespace()
    syncbn_ddp = types.SimpleNamespace(broadcast_buffers=False, process_group=None)
    bcast_ddp = types.SimpleNamespace(broadcast_buffers=True, process_group=None)          # torch DDP's attributes
    one_rank = types.SimpleNamespace(broadcast_buffers=True, process_group="one")
    if ops.OVERLAP_PASSES == "0":
        return
    assert ops.PassOverlap(512, dev, bcast_ddp).on            # no process group initialised: nothing is broadcast
    monkeypatch.setattr(dist, "is_initialized", lambda: True)
    monkeypatch.setattr(dist, "get_world_size", lambda group=None: 1 if group == "one" else 8)
    assert ops.PassOverlap(512, dev, plain).on and ops.PassOverlap(512, dev, syncbn_ddp).on and ops.PassOverlap(16, dev, one_rank).on
    assert not ops.PassOverlap(512, dev, bcast_ddp).on and not ops.PassOverlap(512, dev, plain, bcast_ddp).on
    assert not ops.PassOverlap(512, types.SimpleNamespace(type="cpu", index=None), plain).on


def test_switch_table_reports_live_values(monkeypatch):
    """ssecg.config: every run-time switch is declared once; ``non_default()`` - what bench.py prints as ``config.switches`` - follows
    the LIVE module attributes (a monkeypatched or ``config.set`` value included), and the C library's own per-call switch."""
    from ssecg import amp, config, functional, ops    # noqa: F401  (the owners declare their switches at import)
    snap = config.snapshot()
    for name in ("SSECG_WINOGRAD", "SSECG_WINO_F", "SSECG_WINO_WGRAD_F", "SSECG_WINO_F64", "SSECG_KSPLIT", "SSECG_STEM", "SSECG_STEM_PAIR",
                 "SSECG_OVERLAP_PASSES", "SSECG_BN_MASK_BITS", "SSECG_FUSE_BN", "SSECG_DS_FIRST", "SSECG_AMP_STEM_LP", "SSECG_AMP_STEM_C16",
                 "SSECG_AMP_STEM_BLOCKED", "SSECG_FORCE_SYNC_COLLECTIVES", "SSECG_AMP_WS"):
        assert name in snap, name
    base = config.non_default()
    monkeypatch.setattr(ops, "KSPLIT", not ops.KSPLIT)
    monkeypatch.setenv("SSECG_AMP_WS", "0")
    nd = config.non_default()
    assert ("SSECG_KSPLIT" in nd) != ("SSECG_KSPLIT" in base) and nd["SSECG_AMP_WS"] == "0"
    old = config.get("SSECG_WINO_F")
    config.set("SSECG_WINO_F", 2)
    try:
        assert ops.WINO_F == 2
        with pytest.raises(ValueError):
            config.set("SSECG_WINO_F", 3)
    finally:
        config.set("SSECG_WINO_F", old)
    assert "SSECG_KSPLIT" in config.describe()


def test_bench_counts_executed_multiplications_per_kernel_family():
    """bench.py prices a Winograd kernel against the MFMA peak by the multiplications it EXECUTES: F(4,3) and its transpose (the weight
    gradient) 1/2 of the direct convolution's, F(2,3) and its transpose 2/3, direct kernels all of them."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert b.wino_executed("conv_wino4_kernel<4, 2>") == 0.5
    assert b.wino_executed("conv_wino_wgrad4_kernel + wino_wgrad4_reduce_kernel") == 0.5
    assert abs(b.wino_executed("conv_wino_wgrad_kernel + wino_wgrad_reduce_kernel") - 2.0 / 3.0) < 1e-12
    assert abs(b.wino_executed("conv_wino_kernel<4, 4>") - 2.0 / 3.0) < 1e-12
    assert b.wino_executed("conv_igemm_fast_kernel<256, 128, 4, 2, 3>") == 1.0
    assert b.kernel_class("conv_wino_wgrad4_kernel + wino_wgrad4_reduce_kernel") == "winograd_wgrad"
