"""GPU: the N>1 path end to end with 2 ranks sharing the one MI355X of the test box (gloo carries the collectives,
RCCL refuses two ranks on one device; on the 8-GPU node the same code runs with backend nccl = RCCL).

Checks the §8e equivalence: 2 ranks x B/2 windows with SyncBN + DDP == 1 rank x B windows - same losses, same
gradients (DDP averages per-rank means of equal-size shards), same BN running statistics."""
import os
import time
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FM_FIXTURE = "stepfix_fixmatch_c12_b2_L250"   # the reference's FixMatch step on a batch searched free of ReLU / max-pool / threshold ties
C, B, L, SEED = 12, 2, 2000, 31               # (C, B = the fixture's; L = 2000 for the evaluate() batches)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _setup_paths():
    import sys
    for p in (ROOT, os.path.join(ROOT, "semi-seg-ecg_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _run(rank, world, port, out, backend="gloo", force=False):
    if force:   # one-rank rehearsal: the SyncBN all-reduces are issued although the group has a single rank (read at import)
        os.environ["SSECG_FORCE_SYNC_COLLECTIVES"] = "1"
    _setup_paths()
    import torch.distributed as dist
    from helpers import TRAIN_CFG, build_hip_model, dropout_mask_np, sharpen_for
    from algorithms.base import wrap_ddp
    from algorithms.fixmatch import fixmatch_step
    from ssecg import synth
    dev = torch.device("cuda:0")
    distributed = world > 1 or force
    if distributed:
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        if backend == "nccl":
            torch.cuda.set_device(0)
        dist.init_process_group(backend, rank=rank, world_size=world)
    # the global batch of the reference's tie-free FixMatch fixture, split over the ranks (rows of the labelled and of the
    # unlabelled half, and of the fixed dropout mask): every correct fp32 implementation takes the same ReLU / max-pool /
    # threshold branches on it, so the DDP-averaged SyncBN gradients can be held to the REFERENCE's at 1e-4
    from helpers import StepfixTwin, check_rows, golden
    g = golden(FM_FIXTURE)
    tw = StepfixTwin(g)
    assert (tw.C, tw.B) == (C, B)
    model = build_hip_model(C, tw.sdA_np, dev)
    batch, dm, _ = tw.inputs(0)
    per = B // world
    rows = list(range(rank * per, (rank + 1) * per))
    model.decode_head.fixed_dropout_mask = torch.from_numpy(dm[rows + [B + r for r in rows]]).to(dev, torch.uint8)
    ddp, inner = wrap_ddp({"ddp": {"distributed": distributed, "sync_bn": True, "gpu": 0}}, model)
    from ssecg import functional as SF_
    SF_.COLLECTIVE_LOG = []
    # (the gradient buckets of ssecg.parallel.DataParallel log themselves into the same list, in issue order)
    out["padded_grad_elements"] = sum((p.numel() + 63) // 64 * 64 for p in inner.parameters())
    sl = slice(rank * B // world, (rank + 1) * B // world)
    t = lambda a: torch.from_numpy(a[sl]).to(dev)
    CONF_THRESH = tw.cfg["conf_thresh"]
    loss, stats = fixmatch_step(ddp, t(batch["labeled"]["ecg"]), t(batch["labeled"]["target"]),
                                t(batch["unlabeled"]["ecg"]), t(batch["unlabeled"]["ecg_aug"]), CONF_THRESH)
    loss.backward()
    torch.cuda.synchronize()
    colls, SF_.COLLECTIVE_LOG = SF_.COLLECTIVE_LOG, None
    out[f"colls{rank}"] = colls
    keep_grads = {k: p.grad.detach().cpu().numpy() for k, p in inner.named_parameters() if p.numel() <= 4096 or k.endswith("stem.0.weight")}
    # ALL 65 gradients (DDP has averaged them over the ranks by now) against the reference's, no flip tolerance
    worst_vs_ref = check_rows(g, "step0.grad.", {k: p.grad for k, p in inner.named_parameters()}, 1e-4,
                              what=f"{world}-rank gradients vs the reference")
    keep_bufs = {k: v.detach().cpu().numpy() for k, v in inner.state_dict().items() if "running" in k}
    if distributed:
        s = stats.clone() if backend == "nccl" else stats.clone().cpu()    # RCCL reduces device tensors only
        dist.all_reduce(s)
        stats_mean = (s / world).cpu().numpy()
    else:
        stats_mean = stats.cpu().numpy()
    # evaluate(): every rank scores its shard of each validation batch; only the per-record confusion counts are gathered
    from algorithms.base import evaluate
    vb = [synth.fixmatch_batch(SEED + 5 + i, B, C, L)["labeled"] for i in range(2)]
    loader = [{"ecg": torch.from_numpy(b["ecg"][sl]), "target": torch.from_numpy(b["target"][sl])} for b in vb]
    vstats, vmetrics, vout, vlab = evaluate(ddp, loader, dev, None, use_amp=False)
    if distributed:
        # a second step, only for the collective log: the same buckets in the same order as in the first
        SF_.COLLECTIVE_LOG = []
        ddp.train()
        loss2, _ = fixmatch_step(ddp, t(batch["labeled"]["ecg"]), t(batch["labeled"]["target"]),
                                 t(batch["unlabeled"]["ecg"]), t(batch["unlabeled"]["ecg_aug"]), CONF_THRESH)
        loss2.backward()
        torch.cuda.synchronize()
        out[f"colls_b{rank}"], SF_.COLLECTIVE_LOG = SF_.COLLECTIVE_LOG, None
    if rank == 0:
        out["eval"] = np.array([vstats["loss"], vmetrics["MeanIoU"]])
        out["eval_shapes"] = (tuple(vout.shape), tuple(vlab.shape))
        out["stats"] = stats_mean
        out["worst_vs_ref"] = worst_vs_ref
        out["ref_stats"] = [float(g["step0." + k]) for k in ("loss_total", "loss_x", "loss_u_s", "mask_ratio")]
        out["grads"] = keep_grads
        out["bufs"] = keep_bufs
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


def _join_all(procs, timeout):
    """Join every rank; a rank still alive after ``timeout`` seconds (a hung collective) is terminated and joined BEFORE the
    assertion, so no process is left holding the GPU when the test fails."""
    for p in procs:
        p.join(timeout)
    hung = [p for p in procs if p.is_alive()]
    for p in hung:
        p.terminate()
    for p in hung:
        p.join(30)
        if p.is_alive():
            p.kill()
            p.join(10)
    assert not hung, f"{len(hung)} rank(s) still running after {timeout} s were terminated"
    for p in procs:
        assert p.exitcode == 0, f"rank exited with {p.exitcode}"


def _spawn(world, backend="gloo", force=False):
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_run, args=(r, world, port, out, backend, force)) for r in range(world)]
    for p in procs: p.start()
    _join_all(procs, 300)
    return dict(out)


def test_two_ranks_equal_one_rank():
    one, two = _spawn(1), _spawn(2)
    # the collective SEQUENCE of a FixMatch step (kind, size, dtype in issue order) is identical on both ranks: 18 SyncBN
    # all-reduces in the train-mode forward (the eval-mode teacher pass issues none; a downsample block's two independent
    # BatchNorms share one), 18 in the backward (round 6: the same two share their backward sums too; 21 before), interleaved with
    # the 4 MB gradient buckets - on RCCL a mismatch in order or size would hang or pair the wrong buffers
    assert two["colls0"] == two["colls1"] and len(two["colls0"]) > 0         # first step
    c0, c1 = two["colls_b0"], two["colls_b1"]                                  # second step
    assert c0 == c1 == two["colls0"]
    for cc in (two["colls0"], c0):
        assert sum(1 for c in cc if c[0] == "bn_sums") == 36 and all(c[2] == "torch.float64" for c in cc if c[0] == "bn_sums")
        # the forward's 18: the first convolution and the 1x1 downsample branch of layers 2-4 share one collective (2 x 2C rows)
        fwd = [c[1] for c in cc[:18]]
        assert all(c[0] == "bn_sums" for c in cc[:18]) and max(fwd) == 2 * (512 + 512) and fwd.count(2 * 1024) == 1
        # every parameter gradient, once (each slot of a bucket starts on a 256-byte boundary)
        assert sum(c[1] for c in cc if c[0] == "grad_bucket") == two["padded_grad_elements"] >= 4041284 + 448 * (C - 1)
    buckets = [c for c in c0 if c[0] == "grad_bucket"]
    assert len(buckets) >= 4      # (a bucket closes before it would exceed 4 MB; the 3 MB layer4 weights get one each)
    first_bwd = 18 + next(i for i, c in enumerate(c0[18:]) if c[0] == "grad_bucket")
    assert first_bwd < len(c0) - 5        # gradient buckets start while BatchNorm backward collectives are still being issued
    assert one["colls0"] == []            # single rank: no collective at all
    assert np.allclose(one["stats"], two["stats"], rtol=2e-4, atol=1e-6), (one["stats"], two["stats"])
    assert 0.05 < one["stats"][3] < 0.95, f"mask_ratio {one['stats'][3]}: the masked pseudo-label term must be exercised"
    # both against the reference's real FixMatch step at the global batch: losses 1e-4, every gradient 1e-4 (checked in the ranks)
    for run in (one, two):
        assert np.allclose(run["stats"], run["ref_stats"], rtol=1e-4, atol=1e-6), (run["stats"], run["ref_stats"])
        assert run["worst_vs_ref"] < 1e-4
    print(f"worst gradient statistic vs the reference: 1 rank {one['worst_vs_ref']:.2e}, 2 ranks (DDP + SyncBN) {two['worst_vs_ref']:.2e}")
    for k, v in one["bufs"].items():
        assert np.allclose(v, two["bufs"][k], rtol=1e-5, atol=1e-6), k
    # evaluate(): same loss and MeanIoU whether one rank sees whole batches or two ranks see halves; the returned
    # probabilities / one-hot labels cover all records on every rank, as the reference's all-gather does
    assert np.allclose(one["eval"], two["eval"], rtol=1e-5, atol=1e-7), (one["eval"], two["eval"])
    assert one["eval_shapes"] == two["eval_shapes"] == ((2 * B, 4, L), (2 * B, 4, L))
    worst = 0.0
    ds = {k: float(np.linalg.norm(g - two["grads"][k]) / (np.linalg.norm(g) + 1e-30)) for k, g in one["grads"].items()}
    print("1-rank vs 2-rank gradient differences, largest:", sorted(ds.items(), key=lambda kv: -kv[1])[:6], "smallest:",
          sorted(ds.items(), key=lambda kv: kv[1])[:3])
    for k, d in ds.items():
        worst = max(worst, d)
        assert d < 1e-4, (k, d)  # measured 9e-6: the two runs differ only in the summation order of the BN statistics and gradient averages
    print("worst relative L2 gradient difference 1-rank vs 2-rank:", worst)


def test_rccl_single_rank_rehearsal():
    """RCCL on the one GPU of the test box: a world-size-1 ``nccl`` process group with the SyncBN all-reduces FORCED
    (SSECG_FORCE_SYNC_COLLECTIVES) and the model wrapped in DDP.  Two RCCL ranks cannot share a card, but with one rank every
    collective of the N > 1 step - 36 fp64 BN all-reduces (the backward's 18 async with kernels enqueued before ``work.wait()``) and
    DDP's gradient buckets - goes through ProcessGroupNCCL's own streams, events and tensor bookkeeping around this library's
    raw-stream launches.  The step must equal the reference's (all 65 gradients 1e-4) and the plain single-process run."""
    ref = _spawn(1)
    rccl = _spawn(1, backend="nccl", force=True)
    cc = rccl["colls0"]
    assert sum(1 for c in cc if c[0] == "bn_sums") == 36 and all(c[2] == "torch.float64" for c in cc if c[0] == "bn_sums")
    assert sum(c[1] for c in cc if c[0] == "grad_bucket") == rccl["padded_grad_elements"]
    assert ref["colls0"] == []
    assert rccl["worst_vs_ref"] < 1e-4 and np.allclose(rccl["stats"], rccl["ref_stats"], rtol=1e-4, atol=1e-6)
    assert np.allclose(rccl["stats"], ref["stats"], rtol=1e-6, atol=1e-7)
    for k, v in ref["bufs"].items():
        assert np.allclose(v, rccl["bufs"][k], rtol=1e-6, atol=1e-7), k
    for k, g in ref["grads"].items():
        d = np.linalg.norm(g - rccl["grads"][k]) / (np.linalg.norm(g) + 1e-30)
        assert d < 1e-5, (k, d)       # same kernels, same order: an all-reduce over one rank returns its input
    assert np.allclose(ref["eval"], rccl["eval"], rtol=1e-6, atol=1e-7)
    print(f"one-rank RCCL: {len(cc)} collectives in the step, worst gradient statistic vs the reference {rccl['worst_vs_ref']:.2e}")


# ----------------------------------------------------------------------------------------------------------------------
# MeanTeacher on 2 ranks (BASELINE config #3; src/algorithms/mean_teacher.py:281-319 with the Q3 fix: the frozen teacher is
# NOT wrapped in DDP - the reference's wrap of a module without trainable parameters raises).  Two steps of the plugin's
# real train_one_epoch with the global batch of the reference fixture `mean_teacher_c2_b2` split over the ranks.
MT_FIXTURE = "stepfix_mean_teacher_c2_b2_L250"      # both steps' batches searched tie-free (tools/make_golden.py::gen_step_case)


def _run_mt(rank, world, port, out):
    _setup_paths()
    import torch.distributed as dist
    from helpers import StepfixTwin, build_hip_model, golden
    import algorithms.mean_teacher as A_mt
    from algorithms.base import wrap_ddp
    from ssecg.parallel import DataParallel, unwrap
    from utils.misc import NativeScalerWithGradNormCount
    from utils.optimizer import get_optimizer_from_config
    tw = StepfixTwin(golden(MT_FIXTURE))
    Cm, Bm = tw.C, tw.B
    dev = torch.device("cuda:0")
    distributed = world > 1
    if distributed:
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
    student = build_hip_model(Cm, tw.sdA_np, dev)
    teacher = build_hip_model(Cm, tw.sdB_np, dev)
    for p in teacher.parameters():
        p.requires_grad = False
    with torch.no_grad():
        for pq, pk in zip(student.parameters(), teacher.parameters()):
            pk.data = pq.data                      # mean_teacher.py:285-290 (Q4)
    ddp, inner = wrap_ddp({"ddp": {"distributed": distributed, "sync_bn": True, "gpu": 0}}, student)
    if distributed:
        teacher = torch.nn.SyncBatchNorm.convert_sync_batchnorm(teacher)     # as algorithms/mean_teacher.py:train does
        assert isinstance(ddp, DataParallel)
    assert unwrap(teacher) is teacher and unwrap(ddp) is inner
    cfg = dict(tw.cfg)
    optimizer = get_optimizer_from_config(cfg, inner.parameters())
    scaler = NativeScalerWithGradNormCount()
    calls = {"s": [], "t": []}
    inner.register_forward_hook(lambda m, i, o: calls["s"].append(o["seg_logits"].detach().clone()))
    teacher.register_forward_hook(lambda m, i, o: calls["t"].append(o["seg_logits"].detach().clone()))
    per = Bm // world
    rows = list(range(rank * per, (rank + 1) * per))
    res = {}
    for s in range(2):
        b, dm, _ = tw.inputs(s)
        t = lambda a: torch.from_numpy(a[rows]).to(dev)
        inner.decode_head.fixed_dropout_mask = torch.from_numpy(dm[rows + [Bm + r for r in rows]]).to(dev, torch.uint8)
        calls["s"].clear(); calls["t"].clear()
        stats = A_mt.train_one_epoch(ddp, teacher, [{"ecg": t(b["labeled"]["ecg"]), "target": t(b["labeled"]["target"])}],
                                     [{"ecg": t(b["unlabeled"]["ecg"]), "ecg_aug": t(b["unlabeled"]["ecg_aug"])}], optimizer, dev,
                                     tw.epoch(s), scaler, None, False, cfg)
        res[f"stats{s}"] = {k: float(v) for k, v in stats.items()}
        res[f"logits{s}"] = calls["s"][0].cpu().numpy()
        res[f"pred{s}"] = calls["t"][0].cpu().numpy()
        res[f"teacher{s}"] = {k: v.detach().cpu().numpy() for k, v in teacher.state_dict().items()}
        res[f"student{s}"] = {k: v.detach().cpu().numpy() for k, v in inner.state_dict().items()}
    res["rows"] = rows
    res["scaler"] = scaler.state_dict()
    out[rank] = res
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


def _spawn_mt(world):
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_run_mt, args=(r, world, port, out)) for r in range(world)]
    for p in procs: p.start()
    _join_all(procs, 400)
    return dict(out)


def test_mean_teacher_two_ranks():
    """Two steps of the plugin's real train_one_epoch, the global batch of the reference's tie-free two-step fixture split
    over 2 ranks (1 labelled + 1 unlabelled window each), SyncBN + DDP; teacher unwrapped (Q3)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "semi-seg-ecg_amd"))
    from helpers import StepfixTwin, check_packed, golden
    g = golden(MT_FIXTURE)
    tw = StepfixTwin(g)
    two = _spawn_mt(2)
    one = _spawn_mt(1)[0]
    Bm, decay = tw.B, tw.cfg["ema_decay"]
    r0, r1 = two[0], two[1]
    # (1) the EMA teacher and the student replica are IDENTICAL on both ranks after each step (teacher is not DDP-wrapped:
    # its inputs - DDP-averaged gradients -> identical AdamW updates, SyncBN-synced buffers - are rank-identical)
    for s in range(2):
        for k, v in r0[f"teacher{s}"].items():
            assert np.array_equal(v, r1[f"teacher{s}"][k]), f"teacher {k} differs between ranks after step {s}"
        for k, v in r0[f"student{s}"].items():
            assert np.array_equal(v, r1[f"student{s}"][k]), f"student {k} differs between ranks after step {s}"
    # (2) step 0 against the REFERENCE's fixture generated at the global batch: each rank's logits are its rows of the
    # global-batch logits (SyncBN statistics are global), the rank-averaged losses are the global means
    for r in (r0, r1):
        rows = r["rows"]
        ref_logits = g["step0.logits"][rows + [Bm + i for i in rows]]
        assert np.abs(r["logits0"] - ref_logits).max() < 1e-4 * np.abs(ref_logits).max()
        ref_pred = g["step0.pred_u_w"][rows]
        assert np.abs(r["pred0"] - ref_pred).max() < 1e-4 * np.abs(ref_pred).max()
    for k in ("loss_total", "loss_x", "loss_u_s"):
        ref = float(g["step0." + k])
        assert abs(r0["stats0"][k] - ref) < 1e-4 * max(abs(ref), 1e-3), (k, r0["stats0"][k], ref)
        assert abs(r0["stats0"][k] - r1["stats0"][k]) < 1e-12          # synchronize_between_processes
    lr0 = float(g["step0.lr"])
    assert abs(r0["stats0"]["lr"] - lr0) < 1e-12
    # teacher after the first EMA (Q4: starts as the student) vs the reference's stored update, element by element on the
    # tensors the fixture stores in full, wherever the first AdamW step is well conditioned (|g_ref| > 1e-6 or exactly 0);
    # its buffers (Q5: float32 counters)
    t0 = {k: torch.from_numpy(v) for k, v in r0["teacher0"].items()}
    n_direct = 0
    for k in tw.pnames:
        fk = "step0.tupd.full." + k
        if fk in g.files and ("step0.grad.full." + k) in g.files:
            g0 = np.abs(g["step0.grad.full." + k].astype(np.float64))
            sel = (g0 > 1e-6) | (g0 == 0)
            d = np.abs((t0[k].double().numpy() - tw.sdA_np[k].astype(np.float64)) - g[fk].astype(np.float64))
            if sel.any():
                # (the aliased teacher's first EMA equals the student's whole AdamW step, Q4: no (1 - decay) factor)
                bar = lr0 * np.minimum(2.2, 2e-4 + 1e-4 * np.sqrt((g0 ** 2).mean()) / np.maximum(g0, 1e-300))
                bar = np.where(g0 == 0, 2e-4 * lr0, bar) + 4e-7 * np.abs(tw.sdA_np[k]).max()
                assert (d[sel] <= bar[sel]).all(), (k, (d[sel] / bar[sel]).max())
                n_direct += int(sel.sum())
    assert n_direct > 15000
    check_packed(g, "step0.tbuf.", {k: v for k, v in t0.items() if "running" in k or "num_batches" in k}, 1e-5,
                 what="2-rank teacher buffers step 0")
    check_packed(g, "step0.buf.", {k: torch.from_numpy(v) for k, v in r0["student0"].items() if "running" in k or "num_batches" in k},
                 1e-5, what="2-rank student buffers step 0 (SyncBN statistics are global)")
    assert str(t0["backbone.stem.1.num_batches_tracked"].dtype) == str(g["step0.tbuf.nbt_dtype"])
    # (3) 2 ranks x B/2 == 1 rank x B: same losses, same BN running statistics, same teacher after two steps.  Both are
    # device runs on their own trajectories: after the first (sign-like) AdamW update a handful of elements whose gradient
    # is at the rounding-noise level may have stepped in opposite directions -> counted and bounded, everything else tight.
    for s in range(2):
        for k in ("loss_total", "loss_x", "loss_u_s"):
            assert abs(r0[f"stats{s}"][k] - one[f"stats{s}"][k]) < 2e-4 * max(abs(one[f"stats{s}"][k]), 1e-3), (s, k)
        for k, v in one[f"student{s}"].items():
            if "running" in k:
                # (step 1 follows the first, sign-like AdamW update: the two device runs are on their own trajectories)
                assert np.allclose(v, r0[f"student{s}"][k], rtol=1e-5 if s == 0 else 1e-3, atol=1e-6 if s == 0 else 1e-4), (s, k)
    lr1 = one["stats1"]["lr"]
    n = n_off = 0
    for k, v in one["teacher1"].items():
        d = np.abs(v.astype(np.float64) - r0["teacher1"][k].astype(np.float64))
        if "running" in k or "num_batches" in k:
            assert d.max() <= 1e-5 * max(np.abs(v).max(), 1.0), (k, d.max())
        else:
            # teacher_1 = decay * teacher_0 + (1 - decay) * student_1 and teacher_0 = student_0 (aliased first step, Q4)
            off = d > 2e-3 * lr0 + 4e-7 * np.abs(v).max()
            n += d.size; n_off += int(off.sum())
            assert d.max() <= 2.2 * lr0 + 2.2 * (lr0 + lr1) * (1.0 - decay), (k, d.max())
    assert n_off <= 2e-3 * n, f"{n_off} of {n} teacher elements differ between the 1-rank and the 2-rank run"
    print(f"teacher after 2 steps, 1 rank vs 2 ranks: {n_off} of {n} elements beyond 2e-3 of the first step")
    assert r0["scaler"] == one["scaler"] and r0["scaler"]["_growth_tracker"] == 2 and r0["scaler"]["scale"] == 65536.0


# ----------------------------------------------------------------------------- use_amp (bf16 student pass) under DDP + SyncBN
AMP_C, AMP_B, AMP_SEED = 12, 8, 97


def _amp_inputs():
    """The learnable synthetic batch of tests/test_amp_gpu.py (coherent gradients), the weights, and a threshold at the batch's
    median confidence so that 0 < mask_ratio < 1 (random-init weights pass nothing at 0.8)."""
    _setup_paths()
    from helpers import learnable_batch
    from oracle import torch_ref as O
    from ssecg import synth
    sd_np = synth.model_state(AMP_SEED, AMP_C, trained=True, sharpen=1.0)
    batch = {k: v for k, v in learnable_batch(AMP_SEED + 1, AMP_B, AMP_C, L).items() if k != "u_target"}
    with torch.no_grad():
        conf0 = O.pseudo_label(O.model_forward(O.state_from_numpy(sd_np, requires_grad=False),
                                               torch.from_numpy(batch["unlabeled"]["ecg"]), train=False))[0]
    return sd_np, batch, round(float(conf0.median()), 3)


def _run_amp(rank, world, port, out, amp=True):
    _setup_paths()
    import torch.distributed as dist
    from helpers import build_hip_model
    from algorithms.base import set_amp, wrap_ddp
    from algorithms.fixmatch import fixmatch_step
    from ssecg import functional as SF_
    dev = torch.device("cuda:0")
    distributed = world > 1
    if distributed:
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
    sd_np, batch, thr = _amp_inputs()
    model = build_hip_model(AMP_C, sd_np, dev)
    model.decode_head.dropout = None
    model.decode_head.dropout_ratio = 0.0
    set_amp(amp, model)
    ddp, inner = wrap_ddp({"ddp": {"distributed": distributed, "sync_bn": True, "gpu": 0}}, model)
    SF_.COLLECTIVE_LOG = []
    sl = slice(rank * AMP_B // world, (rank + 1) * AMP_B // world)
    t = lambda a: torch.from_numpy(a[sl]).to(dev)
    loss, stats = fixmatch_step(ddp, t(batch["labeled"]["ecg"]), t(batch["labeled"]["target"]),
                                t(batch["unlabeled"]["ecg"]), t(batch["unlabeled"]["ecg_aug"]), thr)
    loss.backward()
    torch.cuda.synchronize()
    out[f"colls{rank}"] = [c for c in SF_.COLLECTIVE_LOG if c[0] == "bn_sums"]
    s = stats.clone().cpu()
    if distributed:
        dist.all_reduce(s)
        s /= world
    if rank == 0:
        out["stats"] = s.numpy()
        out["grads"] = {k: p.grad.detach().float().cpu().numpy() for k, p in inner.named_parameters()}   # DDP-averaged
        out["bufs"] = {k: v.detach().cpu().numpy() for k, v in inner.state_dict().items() if "running" in k}
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


def _cosd(a, b):
    a, b = np.asarray(a, np.float64).reshape(-1), np.asarray(b, np.float64).reshape(-1)
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))


def test_two_ranks_equal_one_rank_bf16():
    """The reduced-precision student pass (use_amp: true) under DDP + SyncBN: 2 ranks x B/2 against 1 rank x B on a learnable
    batch, judged the way tests/test_amp_gpu.py::test_amp_gradient_cosines_b32 judges the 1-rank step (the
    round-3 bar here was "1.5 x the bf16-vs-fp32 distance"):
    * the collective sequence: 36 fp64 BN all-reduces (round 6: a downsample block's two BatchNorms share their backward sums too), identical on both ranks;
    * losses: 2-rank vs 1-rank <= 2e-3, and each within 5e-3 of the CPU emulation of the precision policy (oracle/amp_ref.py);
    * BN running statistics <= 2e-3;
    * EVERY one of the 65 parameter gradients: cosine(2 ranks, 1 rank) >= 0.95 and >= cosine(1 rank, emulation) - 0.02
      (measured on MI355X, round 5: 0.967-0.9999 against 0.946-0.999).  The two runs differ only in the order in which the
      BatchNorm sums are added (per-rank partials all-reduced vs one set of partial rows): a few bf16 roundings flip and the
      chaotic map does the rest.  Rounds 2-4 measured 0.996-0.9999 (bar 0.98) with ONE rounding per residual unit; since round 5
      the BatchNorm output is rounded before ``out += identity``, as PyTorch's autocast does (tests/test_ampfix_gpu.py), and
      rounded(bn) + identity is EXACTLY zero on ~2e-4 of a block's elements - knife-edge ReLU decisions that any 1-ulp
      perturbation flips, each moving its element's gradient by 100 % (sqrt(2e-4) ~ 1.5 % per block): the reference's own
      numerics under autocast carry the same noise.  A rank that dropped a collective, averaged gradients wrongly or used
      rank-local statistics gives cosines far below 0.9 on the BN parameters."""
    _setup_paths()
    from helpers import TRAIN_CFG, cpu_batch
    from oracle import amp_ref as A
    from oracle import torch_ref as O
    ctx = mp.get_context("spawn")
    res = {}
    for key, world in (("one", 1), ("two", 2)):
        out = ctx.Manager().dict()
        port = _free_port()
        procs = [ctx.Process(target=_run_amp, args=(r, world, port, out, True)) for r in range(world)]
        for p in procs: p.start()
        _join_all(procs, 300)
        res[key] = dict(out)
    one, two = res["one"], res["two"]
    assert two["colls0"] == two["colls1"] and len(two["colls0"]) == 36 and one["colls0"] == []
    sd_np, batch, thr = _amp_inputs()
    cfg = dict(TRAIN_CFG, conf_thresh=thr, betas=(0.9, 0.999))
    remu = A.fixmatch_step(O.state_from_numpy(sd_np), {}, cpu_batch(batch), cfg, 3.0, None)
    emu_stats = np.array([remu["loss_total"], remu["loss_x"], remu["loss_u_s"], remu["mask_ratio"]])
    assert 0.1 < emu_stats[3] < 0.9
    for j, k in enumerate(("loss_total", "loss_x", "loss_u_s")):
        assert abs(one["stats"][j] - two["stats"][j]) <= 2e-3 * max(abs(one["stats"][j]), 1e-3), (k, one["stats"], two["stats"])
        assert abs(one["stats"][j] - emu_stats[j]) <= 5e-3 * max(abs(emu_stats[j]), 1e-3), (k, one["stats"], emu_stats)
    assert abs(one["stats"][3] - two["stats"][3]) <= 2e-3
    for k, v in one["bufs"].items():
        assert np.allclose(v, two["bufs"][k], rtol=2e-3, atol=2e-4), k
    rows = [(k, _cosd(two["grads"][k], one["grads"][k]), _cosd(one["grads"][k], remu["grads"][k].detach().numpy())) for k in one["grads"]]
    print("cosine (2 ranks vs 1 rank) and (1 rank vs emulation), eight lowest:")
    for k, c21, c1e in sorted(rows, key=lambda t: t[1])[:8]:
        print(f"  {k:45s} {c21:.4f} {c1e:.4f}")
    for k, c21, c1e in rows:
        assert c21 >= 0.95, f"{k}: 2-rank vs 1-rank cosine {c21:.4f}"
        assert c21 >= c1e - 0.02, f"{k}: 2-rank vs 1-rank cosine {c21:.4f} < 1-rank-vs-emulation {c1e:.4f} - 0.02"
        assert c1e >= 0.93, f"{k}: 1-rank HIP vs emulation cosine {c1e:.4f}"


# ----------------------------------------------------------------------------------------------------------------------
# The whole step as ONE HIP graph under torch.distributed (round 6): over RCCL the collectives are captured with the step.
def _run_graph(rank, world, port, out, graph):
    os.environ["SSECG_FORCE_SYNC_COLLECTIVES"] = "1"
    _setup_paths()
    import torch.distributed as dist
    import utils.lr_sched as lr_sched
    from helpers import TRAIN_CFG, build_hip_model
    from algorithms.base import dist_graph_ok, wrap_ddp
    from algorithms.fixmatch import fixmatch_step
    from ssecg import functional as SF_
    from ssecg import synth
    from ssecg.graph import StepGraph
    from ssecg.parallel import DataParallel
    from utils.misc import NativeScalerWithGradNormCount
    from utils.optimizer import get_optimizer_from_config
    dev = torch.device("cuda:0")
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    Cg, Bg, Lg, n = 2, 4, 500, 7
    model = build_hip_model(Cg, synth.model_state(5, Cg, trained=True), dev)
    ddp, inner = wrap_ddp({"ddp": {"distributed": True, "sync_bn": True, "gpu": 0}}, model)
    assert isinstance(ddp, DataParallel) and dist_graph_ok(ddp)
    cfg = dict(TRAIN_CFG)
    opt = get_optimizer_from_config(cfg, inner.parameters())
    scaler = NativeScalerWithGradNormCount()
    torch.manual_seed(1234)

    def whole_step(ecg_x, mask_x, ecg_u_w, ecg_u_s):
        loss, stats = fixmatch_step(ddp, ecg_x, mask_x, ecg_u_w, ecg_u_s, 0.3)
        scaler(loss, opt, clip_grad=None, parameters=ddp.parameters(), update_grad=True)
        opt.zero_grad()
        return stats

    step = StepGraph(whole_step) if graph else whole_step
    stats, per_step = [], []
    for i in range(n):
        b = synth.fixmatch_batch(100 + i, Bg, Cg, Lg)
        t = lambda a: torch.from_numpy(a).to(dev)
        lr_sched.adjust_learning_rate(opt, 3.0 + i / 7.0, cfg)
        SF_.COLLECTIVE_LOG = []
        t_call = time.perf_counter()
        stats.append(step(t(b["labeled"]["ecg"]), t(b["labeled"]["target"]), t(b["unlabeled"]["ecg"]), t(b["unlabeled"]["ecg_aug"])).clone())
        if i == 2:
            out["third_call_s"] = time.perf_counter() - t_call     # the capturing call (two eager warm-up calls precede it)
        per_step.append(len(SF_.COLLECTIVE_LOG))
        SF_.COLLECTIVE_LOG = None
    torch.cuda.synchronize()
    out["stats"] = torch.stack(stats).cpu().numpy()
    out["state"] = {k: v.detach().cpu().numpy() for k, v in inner.state_dict().items()}
    out["moments"] = [v["exp_avg"].detach().cpu().numpy() for v in opt.state_dict()["state"].values()]
    out["per_step"] = per_step
    if graph:
        out["replays"] = step.replays
        out["captured"] = step.graph is not None
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_single_rank_step_graph_is_bit_identical():
    """A world-size-1 ``nccl`` group with the SyncBN all-reduces forced and the model in this library's DataParallel: the FixMatch
    step captured into ONE HIP graph together with its 36 fp64 SyncBN all-reduces and its gradient buckets (ProcessGroupNCCL
    launches on RCCL's stream, their waits the graph's edges) and replayed must equal the same steps run eagerly through the same
    collectives - statistics, weights, BatchNorm buffers, AdamW moments - bit for bit; the eager steps and the capture issue the
    same number of collectives, a replay issues none from the host."""
    def spawn(graph):
        ctx = mp.get_context("spawn")
        out = ctx.Manager().dict()
        p = ctx.Process(target=_run_graph, args=(0, 1, _free_port(), out, graph))
        p.start()
        _join_all([p], 300)
        return dict(out)

    eager, graphed = spawn(False), spawn(True)
    assert graphed["captured"] and graphed["replays"] == 5, graphed.get("replays")
    # the capture under RCCL first lets ProcessGroupNCCL's watchdog drop the eager steps' works (ssecg.graph.NCCL_WATCHDOG_DRAIN_S): without
    # it 3 of 60 captures of exactly this step ended the process with hipErrorCapturedEvent (profiles/r06_rccl_capture_watchdog.txt)
    from ssecg.graph import NCCL_WATCHDOG_DRAIN_S
    assert NCCL_WATCHDOG_DRAIN_S >= 0.25 and graphed["third_call_s"] >= NCCL_WATCHDOG_DRAIN_S > eager["third_call_s"]
    assert len(set(eager["per_step"])) == 1 and eager["per_step"][0] >= 37            # 36 SyncBN + the gradient buckets, every step
    assert graphed["per_step"][:3] == eager["per_step"][:3] and set(graphed["per_step"][3:]) == {0}
    assert np.array_equal(eager["stats"], graphed["stats"])
    for k, v in eager["state"].items():
        assert np.array_equal(v, graphed["state"][k]), k
    for a, b in zip(eager["moments"], graphed["moments"]):
        assert np.array_equal(a, b)
