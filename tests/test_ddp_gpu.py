"""GPU: the N>1 path end to end with 2 ranks sharing the one MI355X of the test box (gloo carries the collectives,
RCCL refuses two ranks on one device; on the 8-GPU node the same code runs with backend nccl = RCCL).

Checks the §8e equivalence: 2 ranks x B/2 windows with SyncBN + DDP == 1 rank x B windows - same losses, same
gradients (DDP averages per-rank means of equal-size shards), same BN running statistics."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C, B, L, SEED = 2, 4, 2000, 31


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _setup_paths():
    import sys
    for p in (ROOT, os.path.join(ROOT, "semi-seg-ecg_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _run(rank, world, port, out):
    _setup_paths()
    import torch.distributed as dist
    from helpers import TRAIN_CFG, build_hip_model, dropout_mask_np, sharpen_for
    from algorithms.base import wrap_ddp
    from algorithms.fixmatch import fixmatch_step
    from ssecg import synth
    dev = torch.device("cuda:0")
    distributed = world > 1
    if distributed:
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
    model = build_hip_model(C, synth.model_state(SEED, C, trained=True, sharpen=sharpen_for(C)), dev)
    model.decode_head.dropout = None  # dropout off: the per-rank masks would differ from the single-rank one
    model.decode_head.dropout_ratio = 0.0
    ddp, inner = wrap_ddp({"ddp": {"distributed": distributed, "sync_bn": True, "gpu": 0}}, model)
    batch = synth.fixmatch_batch(SEED + 1, B, C, L)
    sl = slice(rank * B // world, (rank + 1) * B // world)
    t = lambda a: torch.from_numpy(a[sl]).to(dev)
    loss, stats = fixmatch_step(ddp, t(batch["labeled"]["ecg"]), t(batch["labeled"]["target"]),
                                t(batch["unlabeled"]["ecg"]), t(batch["unlabeled"]["ecg_aug"]), TRAIN_CFG["conf_thresh"])
    loss.backward()
    from ssecg.functional import wait_for_wgrads
    wait_for_wgrads()
    torch.cuda.synchronize()
    if distributed:
        s = stats.clone().cpu()
        dist.all_reduce(s)
        stats_mean = (s / world).numpy()
    else:
        stats_mean = stats.cpu().numpy()
    # evaluate(): every rank scores its shard of each validation batch; only the per-record confusion counts are gathered
    from algorithms.base import evaluate
    vb = [synth.fixmatch_batch(SEED + 5 + i, B, C, L)["labeled"] for i in range(2)]
    loader = [{"ecg": torch.from_numpy(b["ecg"][sl]), "target": torch.from_numpy(b["target"][sl])} for b in vb]
    vstats, vmetrics, vout, vlab = evaluate(ddp, loader, dev, None, use_amp=False)
    if rank == 0:
        out["eval"] = np.array([vstats["loss"], vmetrics["MeanIoU"]])
        out["eval_shapes"] = (tuple(vout.shape), tuple(vlab.shape))
        out["stats"] = stats_mean
        out["grads"] = {k: p.grad.detach().cpu().numpy() for k, p in inner.named_parameters()
                        if p.numel() <= 4096 or k.endswith("stem.0.weight")}
        out["bufs"] = {k: v.detach().cpu().numpy() for k, v in inner.state_dict().items() if "running" in k}
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


def _spawn(world):
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_run, args=(r, world, port, out)) for r in range(world)]
    for p in procs: p.start()
    for p in procs: p.join(300)
    for p in procs:
        assert p.exitcode == 0, f"rank exited with {p.exitcode}"
    return dict(out)


def test_two_ranks_equal_one_rank():
    one, two = _spawn(1), _spawn(2)
    assert np.allclose(one["stats"], two["stats"], rtol=2e-4, atol=1e-6), (one["stats"], two["stats"])
    for k, v in one["bufs"].items():
        assert np.allclose(v, two["bufs"][k], rtol=1e-5, atol=1e-6), k
    # evaluate(): same loss and MeanIoU whether one rank sees whole batches or two ranks see halves; the returned
    # probabilities / one-hot labels cover all records on every rank, as the reference's all-gather does
    assert np.allclose(one["eval"], two["eval"], rtol=1e-5, atol=1e-7), (one["eval"], two["eval"])
    assert one["eval_shapes"] == two["eval_shapes"] == ((2 * B, 4, L), (2 * B, 4, L))
    worst = 0.0
    for k, g in one["grads"].items():
        d = np.linalg.norm(g - two["grads"][k]) / (np.linalg.norm(g) + 1e-30)
        worst = max(worst, d)
        assert d < 2e-2, (k, d)  # flip-tolerant (see tests/helpers.py); typically ~1e-5
    print("worst relative L2 gradient difference 1-rank vs 2-rank:", worst)
