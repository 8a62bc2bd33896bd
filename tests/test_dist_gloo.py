"""CPU, world_size 2 over gloo: the data-parallel plumbing of the hot path - rank seeding / sharding of the
synthetic windows, SyncBN statistic merging (all-reduced fp64 sums with global count = local * world),
packed metric all-reduce, DDP wrapping (gradient averaging).  The kernels themselves need a GPU; what is
checked here is everything the N>1 path adds around them."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, ret):
    import sys
    for p in (ROOT, os.path.join(ROOT, "semi-seg-ecg_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import utils.misc as misc
    from ssecg import functional as SF
    from ssecg import synth
    cfg = {"dist_url": "env://", "dist_backend": "gloo"}
    misc.init_distributed_mode(cfg, with_time=False)
    assert cfg["distributed"] and misc.get_world_size() == world and misc.get_rank() == rank
    # (1) SyncBN merge: each rank holds half of a global batch; all-reduced sums == global sums
    x = torch.from_numpy(synth.normal(5, 1, (4, 8, 50))).double()
    mine = x[rank * 2:(rank + 1) * 2]
    sums = torch.stack([mine.sum(dim=(0, 2)), (mine ** 2).sum(dim=(0, 2))], dim=1).contiguous()
    sums = SF._allreduce_sums(sums, dist.group.WORLD)
    count = mine.shape[0] * mine.shape[2] * world
    mean = sums[:, 0] / count
    var = sums[:, 1] / count - mean ** 2
    assert torch.allclose(mean, x.mean(dim=(0, 2)), atol=1e-12) and torch.allclose(var, x.var(dim=(0, 2), unbiased=False), atol=1e-12)
    # (1b) the async form used to overlap a BatchNorm's collective with the other branch's conv: issued in program order,
    # waited in any order, same sums; the issue log is what the 2-rank GPU test compares between ranks
    SF.COLLECTIVE_LOG = []
    a = torch.full((8, 2), float(rank + 1), dtype=torch.float64)
    b = torch.full((16, 2), float(10 * (rank + 1)), dtype=torch.float64)
    ta, wa = SF._allreduce_sums_async(a, dist.group.WORLD)
    tb, wb = SF._allreduce_sums_async(b, dist.group.WORLD)
    wb.wait(); wa.wait()
    assert torch.equal(ta, torch.full((8, 2), 3.0, dtype=torch.float64)) and torch.equal(tb, torch.full((16, 2), 30.0, dtype=torch.float64))
    logs = [None, None]
    dist.all_gather_object(logs, SF.COLLECTIVE_LOG)
    assert logs[0] == logs[1] == [("bn_sums", 16, "torch.float64"), ("bn_sums", 32, "torch.float64")]
    SF.COLLECTIVE_LOG = None
    # (2) BNState picks the process group up from a converted SyncBatchNorm
    bn = torch.nn.SyncBatchNorm.convert_sync_batchnorm(torch.nn.Sequential(torch.nn.BatchNorm1d(8)))[0]
    assert SF.BNState.of(bn).group is not None and SF.BNState.of(torch.nn.BatchNorm1d(8)).group is None
    # (3) packed metric all-reduce: one collective for all rows and names
    ml = misc.MetricLogger()
    buf = misc.DeviceMetricBuffer(["a", "b"], 3, torch.device("cpu"))
    for i in range(3):
        buf.push(torch.tensor([float(rank + i), 10.0 * rank]))
    rows = buf.flush(ml)
    assert rows[2] == {"a": 0.5 + 2, "b": 5.0} and ml.meters["a"].global_avg == rank + 1.0
    assert misc.all_reduce_mean(float(rank)) == 0.5
    # (4) DDP wrap: gradients are averaged over ranks
    from algorithms.base import wrap_ddp
    torch.manual_seed(0)
    m = torch.nn.Linear(4, 1)
    ddp, inner = wrap_ddp({"ddp": {"distributed": True, "sync_bn": True, "gpu": rank}}, m)
    ddp(torch.full((2, 4), float(rank + 1))).sum().backward()
    assert torch.allclose(inner.weight.grad, torch.full((1, 4), 2.0 * 1.5))
    from ssecg.parallel import DataParallel as _DP
    assert isinstance(ddp, _DP)
    # ``ddp.reducer: torch`` keeps torch's wrapper (the comparison switch): same averaged gradient
    torch.manual_seed(0)
    m_t = torch.nn.Linear(4, 1)
    ddp_t, inner_t = wrap_ddp({"ddp": {"distributed": True, "sync_bn": True, "gpu": rank, "reducer": "torch"}}, m_t)
    assert isinstance(ddp_t, torch.nn.parallel.DistributedDataParallel)
    ddp_t(torch.full((2, 4), float(rank + 1))).sum().backward()
    assert torch.allclose(inner_t.weight.grad, inner.weight.grad)
    # (5) per-rank data: seeds differ by rank, so shards differ
    a = synth.fixmatch_batch(1234 + rank, 2, 1, 200)["labeled"]["ecg"]
    g = [torch.zeros(2, 1, 200) for _ in range(world)]
    dist.all_gather(g, torch.from_numpy(a))
    assert not torch.equal(g[0], g[1])
    # (7) a plugin epoch with a TensorBoard writer on rank 0 ONLY (as output_dir_and_writer gives it): the packed metric
    # all-reduce must still be issued by both ranks, in step with the DDP gradient buckets (a one-sided
    # all-reduce pairs with the other rank's next collective).  CPU stand-in model: the collectives are what is tested.
    import algorithms.base as A_base

    class _Stub(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = torch.nn.Linear(6, 3)

        def forward(self, inputs, labels=None, return_loss=False):
            lo = self.lin(inputs.mean(dim=2))
            return {"seg_logits": lo, "loss": torch.nn.functional.cross_entropy(lo, labels[:, 0])}

    class _Writer:
        def __init__(self): self.rows = []
        def add_scalar(self, k, v, x): self.rows.append((k, v, x))

    def _scaler(loss, optimizer, clip_grad=None, parameters=None, update_grad=True):
        loss.backward()
        if update_grad:
            optimizer.step()

    torch.manual_seed(3)
    stub, _ = wrap_ddp({"ddp": {"distributed": True, "sync_bn": False, "gpu": rank}}, _Stub())
    gen = torch.Generator().manual_seed(10 + rank)
    loader = [{"ecg": torch.randn(4, 6, 20, generator=gen), "target": torch.randint(0, 3, (4, 20), generator=gen)} for _ in range(45)]
    writer = _Writer() if rank == 0 else None
    stats = A_base.train_one_epoch(stub, loader, torch.optim.SGD(stub.parameters(), lr=0.1), torch.device("cpu"), 0, _scaler,
                                   writer, use_amp=False, config=dict(accum_iter=1, max_norm=None, warmup_epochs=1, lr=0.1,
                                                                     min_lr=0.0, epochs=2))
    both = [None, None]
    dist.all_gather_object(both, stats["loss"])
    assert abs(both[0] - both[1]) < 1e-9          # global averages are synchronised between processes
    if rank == 0:
        logged = [r for r in writer.rows if r[0] == "loss"]
        assert len(logged) == 45                    # rank 0 logged the RANK-AVERAGED per-step values
    w = [torch.zeros_like(stub.module.lin.weight) for _ in range(world)]
    dist.all_gather(w, stub.module.lin.weight.detach())
    assert torch.equal(w[0], w[1])                   # and the DDP replicas stayed identical (no mispaired collective)
    # (8) the reducer of ssecg.parallel.DataParallel against torch's DistributedDataParallel on the same replicas: several
    # buckets, a parameter that gets no gradient on one rank, accumulation over two micro-steps WITHOUT no_sync (the reference's
    # accum_iter: every backward all-reduces, src/algorithms/fixmatch.py:73-78,129-138), then with no_sync; the gradient
    # collectives are issued in the same order and sizes on both ranks; ``.grad`` lives inside the bucket afterwards
    from ssecg.parallel import DataParallel, unwrap

    def _net():
        torch.manual_seed(11)
        return torch.nn.Sequential(torch.nn.Linear(6, 40), torch.nn.ReLU(), torch.nn.Linear(40, 30), torch.nn.ReLU(), torch.nn.Linear(30, 2))

    class _Branchy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.net, self.side = _net(), torch.nn.Linear(6, 2)

        def forward(self, x, use_side):
            return self.net(x) + (self.side(x) if use_side else 0.0)

    mine_m, ref_m = _Branchy(), _Branchy()
    mine = DataParallel(mine_m, bucket_cap_mb=0.002, broadcast_buffers=False)     # 524 floats per bucket -> 3 buckets
    ref = torch.nn.parallel.DistributedDataParallel(ref_m, find_unused_parameters=True)
    assert len(mine._buckets) == 3 and unwrap(mine) is mine_m and unwrap(ref) is ref_m and unwrap(mine_m) is mine_m
    g2 = torch.Generator().manual_seed(50 + rank)
    SF.COLLECTIVE_LOG = []
    for it in range(3):
        x = torch.randn(5, 6, generator=g2)
        side = (rank == 0) if it == 1 else (it == 0)       # step 1: only rank 0 uses the side branch; step 2: nobody does
        for m in (mine, ref):
            m.zero_grad(set_to_none=True)
            m(x, side).square().sum().backward()
        for (n, a), b in zip(mine_m.named_parameters(), ref_m.parameters()):
            if b.grad is None:
                assert a.grad is None or not a.grad.any(), n
            else:
                assert a.grad is not None and torch.allclose(a.grad, b.grad, rtol=1e-6, atol=1e-7), (it, n)
                bi, i = mine._slot[a]
                assert a.grad.data_ptr() == mine._buckets[bi].view(i).data_ptr()
    logs = [None, None]
    dist.all_gather_object(logs, SF.COLLECTIVE_LOG)
    assert logs[0] == logs[1] and len(logs[0]) == 3 * len(mine._buckets) and {c[0] for c in logs[0]} == {"grad_bucket"}
    SF.COLLECTIVE_LOG = None
    # accumulation, every micro-step reduced (grads are not reset in between): avg(g1) + avg(g2)
    for m in (mine, ref):
        m.zero_grad(set_to_none=True)
    for k in range(2):
        x = torch.randn(5, 6, generator=g2)
        for m in (mine, ref):
            (m(x, True).square().sum() / 2).backward()
    for a, b in zip(mine_m.parameters(), ref_m.parameters()):
        assert torch.allclose(a.grad, b.grad, rtol=1e-6, atol=1e-7)
    # no_sync: the first micro-step stays local, the second reduces the sum
    for m in (mine, ref):
        m.zero_grad(set_to_none=True)
    xs = [torch.randn(5, 6, generator=g2) for _ in range(2)]
    for m in (mine, ref):
        with m.no_sync():
            m(xs[0], True).square().sum().backward()
        m(xs[1], True).square().sum().backward()
    for a, b in zip(mine_m.parameters(), ref_m.parameters()):
        assert torch.allclose(a.grad, b.grad, rtol=1e-6, atol=1e-7)
    # wrapping the same module again replaces the first wrapper's hooks (two reducers would scale the gradients twice)
    again = DataParallel(mine_m, bucket_cap_mb=0.002, broadcast_buffers=False)
    assert mine._hooks == [] and len(again._hooks) == len(list(mine_m.parameters()))
    again.zero_grad(set_to_none=True); ref.zero_grad(set_to_none=True)
    for m in (again, ref):
        m(xs[0], True).square().sum().backward()
    for a, b in zip(mine_m.parameters(), ref_m.parameters()):
        assert torch.allclose(a.grad, b.grad, rtol=1e-6, atol=1e-7)
    # construction broadcast rank 0's parameters and buffers
    torch.manual_seed(100 + rank)
    bnm = torch.nn.Sequential(torch.nn.Linear(3, 3), torch.nn.BatchNorm1d(3))
    bnm[1].running_mean.fill_(float(rank + 1))
    DataParallel(bnm)
    got = [torch.zeros(3 * 3 + 3) for _ in range(world)]
    dist.all_gather(got, torch.cat([bnm[0].weight.detach().flatten(), bnm[1].running_mean]))
    assert torch.equal(got[0], got[1]) and got[1][-1] == 1.0
    # (6) meters synchronise
    ml.synchronize_between_processes()
    assert ml.meters["a"].count == 6
    dist.barrier()
    dist.destroy_process_group()
    ret[rank] = True


def test_world_size_2_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs: p.start()
    for p in procs: p.join(180)
    for p in procs:
        assert p.exitcode == 0, "a rank failed"
    assert dict(ret) == {0: True, 1: True}
