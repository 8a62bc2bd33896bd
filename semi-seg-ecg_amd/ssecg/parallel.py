"""The data-parallel wrapper of the N > 1 step: one process per GPU, gradients averaged over the ranks by RCCL.

Replaces ``torch.nn.parallel.DistributedDataParallel`` where the reference wraps its models
(``src/algorithms/fixmatch.py:292-295``, ``mean_teacher.py:310-318``, ``base.py:325-328``, ``cps.py:298-305``,
``stpp.py:325-328,613-621``) with the same contract - parameters and buffers of rank 0 broadcast at construction,
``.module``, gradients averaged over the process group by the end of ``backward()``, buffers re-broadcast before a forward
when asked, ``no_sync()`` - and a reducer shaped for this library's step instead of torch's general one:

* the gradients of a bucket are staged into the bucket's flat fp32 buffer, scaled by 1 / world, by ONE multi-tensor launch
  (``ssecg_pack_scaled_multi``); torch's reducer issues one ``mul_out`` per parameter (65 per step, measured +0.39 ms on
  one rank, ``profiles/r04_dist_overhead_one_rank.txt``);
* ``param.grad`` is then re-pointed at its slice of the flat buffer, so the all-reduce result is read in place by the
  multi-tensor gradient-norm / AdamW kernels, whose pointer tables stop changing from step to step;
* buckets are filled in reverse registration order (the order the backward produces gradients in) and their all-reduces
  are issued in bucket order on every rank, as soon as the last gradient of a bucket has been accumulated, overlapping the
  ring all-reduce over xGMI with the rest of the backward; the end-of-backward callback waits for them.

A parameter that received no gradient on this rank contributes zeros and gets the averaged gradient of the others (torch
DDP with ``find_unused_parameters=True``; one that no rank used ends with a zero gradient rather than ``None`` - the
reference's default ``find_unused_parameters=False`` raises in either case, so no run of the reference depends on it).
That holds only while the gradient buckets are the ONLY collectives of the backward pass: a rank that misses a gradient
holds its bucket (and every later one) back to the end of the backward, the other ranks issue theirs in between their
SyncBatchNorm all-reduces, and the per-rank collective ORDER on the shared process group would differ (a hang, or an fp32
bucket paired with another rank's fp64 BatchNorm sums).  With SyncBatchNorm collectives in flight such a backward raises
``SsecgError`` instead - the reference's own DDP raises for ANY unused parameter.

Gradient accumulation without ``no_sync`` (what the reference does: every micro-step's backward all-reduces) keeps DDP's
arithmetic: the accumulated ``.grad`` - the flat slice itself - is scaled and summed again, avg(g1) + avg(g2).
"""
from __future__ import annotations

import contextlib
import weakref

import torch
import torch.distributed as dist
from torch.autograd import Variable

from . import functional as SF
from . import ops
from .lib import SsecgError

_ALIGN = 64   # elements: every slot starts on a 256-byte boundary of its bucket
#: how long a forward waits for the collectives a RAISED backward left in flight (``_rearm``) before it gives the job up
REARM_TIMEOUT_S = 20.0


class _Bucket:
    __slots__ = ("flat", "params", "offsets", "pending", "work", "table", "launched")

    def __init__(self, params, device):
        self.params, self.offsets, n = params, [], 0
        for p in params:
            self.offsets.append(n)
            n += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.flat = torch.zeros(n, dtype=torch.float32, device=device)
        self.pending, self.work, self.table, self.launched = len(params), None, {}, False

    def view(self, i):
        p = self.params[i]
        return self.flat[self.offsets[i]:self.offsets[i] + p.numel()].view(p.shape)


class DataParallel(torch.nn.Module):
    #: test hook: ``f(params, views, scale)`` that stages host gradients into the bucket (the CPU / gloo rehearsals of the
    #: reducer's bookkeeping install one); None in the product: host tensors raise
    HOST_STAGER = None

    def __init__(self, module, process_group=None, bucket_cap_mb=4.0, broadcast_buffers=True):
        super().__init__()
        if not dist.is_initialized():
            raise SsecgError("DataParallel needs an initialised process group")
        self.module = module
        self.process_group = process_group if process_group is not None else dist.group.WORLD
        self.world_size = dist.get_world_size(self.process_group)
        self.broadcast_buffers = bool(broadcast_buffers)
        self.require_backward_grad_sync = True
        # a module that is wrapped again (a plugin that re-enters its setup) must not keep the previous wrapper's hooks: two
        # reducers would each scale and all-reduce the same gradients
        prev = getattr(module, "_ssecg_dp_wrapper", None)
        prev = prev() if prev is not None else None
        if prev is not None:
            prev.remove_hooks()
        object.__setattr__(module, "_ssecg_dp_wrapper", weakref.ref(self))
        self._hooks = []
        params = [p for p in module.parameters() if p.requires_grad]
        for p in params:
            if p.dtype != torch.float32 or p.device != params[0].device:
                raise SsecgError("DataParallel: parameters must be fp32 tensors on one device")
        self._sync_module_states()
        self._buckets, self._slot = [], {}
        cap = max(int(bucket_cap_mb * (1 << 20)) // 4, 1)
        cur, size = [], 0
        for p in reversed(params):
            if cur and size + p.numel() > cap:
                self._buckets.append(_Bucket(cur, p.device))
                cur, size = [], 0
            cur.append(p)
            size += p.numel()
        if cur:
            self._buckets.append(_Bucket(cur, cur[0].device))
        for bi, b in enumerate(self._buckets):
            for i, p in enumerate(b.params):
                self._slot[p] = (bi, i)
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self._armed = False
        self._next = 0
        self._bn_seen = 0

    def remove_hooks(self):
        """Detach this wrapper's gradient hooks (it stops reducing; ``.module`` stays usable)."""
        for h in self._hooks:
            h.remove()
        self._hooks = []

    # ------------------------------------------------------------------ state synchronisation
    def _broadcast(self, tensors):
        """Rank 0's values into every rank's tensors: ONE flat buffer per dtype (fp32 parameters / statistics, the int64
        ``num_batches_tracked`` counters), one ``dist.broadcast`` each, copied back in place - public API only."""
        tensors = [t for t in tensors if t is not None and t.numel() > 0]
        if not tensors or self.world_size <= 1:
            return
        src = dist.get_global_rank(self.process_group, 0)
        by_dtype = {}
        for t in tensors:
            by_dtype.setdefault((t.dtype, t.device), []).append(t)
        for (dtype, device), ts in by_dtype.items():
            flat = torch.cat([t.detach().reshape(-1) for t in ts])
            dist.broadcast(flat, src=src, group=self.process_group)
            off = 0
            for t in ts:
                n = t.numel()
                t.detach().copy_(flat[off:off + n].view(t.shape))
                off += n

    def _sync_module_states(self):
        with torch.no_grad():
            self._broadcast([p.detach() for p in self.module.parameters()] + list(self.module.buffers()))

    def forward(self, *args, **kwargs):
        if torch.is_grad_enabled() and self.require_backward_grad_sync:
            self._rearm()
        if self.broadcast_buffers and self.require_backward_grad_sync:
            with torch.no_grad():
                self._broadcast(list(self.module.buffers()))
        return self.module(*args, **kwargs)

    def _rearm(self):
        """Start of a forward whose backward will reduce (torch's ``prepare_for_backward``).  The reducer's state is normally
        reset by the end-of-backward callback; a backward that RAISED (an out-of-memory retry, a skipped bad batch) never ran
        it - buckets half counted down, collectives started and never waited for.  Wait for what was started and count from
        zero again - but only for ``REARM_TIMEOUT_S``: the wait completes when EVERY rank started the same collectives
        up to the failure (a bad batch that every rank skips).  A RANK-LOCAL failure (one rank out of memory) leaves the others
        without a partner; recovering from that is unsupported - this rank then fails fast with ``SsecgError`` instead of sitting
        in the wait until the process group's own timeout (600 s)."""
        if self._armed or self._next or any(b.work is not None or b.pending != len(b.params) for b in self._buckets):
            import datetime
            for b in self._buckets:
                if b.work is not None:
                    try:
                        done = b.work.wait(datetime.timedelta(seconds=REARM_TIMEOUT_S))
                    except Exception as e:  # noqa: BLE001 - the backends raise their own timeout types
                        done, err = False, e
                    else:
                        err = None
                    if done is False:
                        raise SsecgError(
                            f"DataParallel: a gradient bucket started by a backward pass that raised did not complete within "
                            f"{REARM_TIMEOUT_S:.0f} s - the other ranks never issued its partner (a rank-local failure).  Recovering "
                            f"from that is unsupported: the process group holds an unmatched collective, this process must exit.") from err
            for b in self._buckets:
                b.pending, b.work, b.launched = len(b.params), None, False
            self._armed, self._next = False, 0

    @contextlib.contextmanager
    def no_sync(self):
        """Gradients of backward passes inside the context stay local (accumulate into ``.grad``); the first backward outside it
        reduces the accumulated sums (``DistributedDataParallel.no_sync``)."""
        old, self.require_backward_grad_sync = self.require_backward_grad_sync, False
        try:
            yield
        finally:
            self.require_backward_grad_sync = old

    # ------------------------------------------------------------------ reducer
    def _on_grad(self, p):
        if not self.require_backward_grad_sync:
            return
        if not self._armed:
            self._armed = True
            self._bn_seen = SF.COLLECTIVES_ISSUED[0]
            Variable._execution_engine.queue_callback(self._finish)
        b = self._buckets[self._slot[p][0]]
        b.pending -= 1
        if b.pending == 0:
            self._launch_ready()

    def _launch_ready(self):
        while self._next < len(self._buckets) and self._buckets[self._next].pending <= 0:
            self._launch(self._buckets[self._next])
            self._next += 1

    def _launch(self, b):
        scale = 1.0 / self.world_size
        rows, mx = [], 0
        views = [b.view(i) for i in range(len(b.params))]
        for i, p in enumerate(b.params):
            g = p.grad
            if g is not None and (g.dtype != torch.float32 or not g.is_contiguous() or g.device != b.flat.device):
                raise SsecgError("DataParallel: gradients must be contiguous fp32 tensors on the parameters' device")
            rows += [0 if g is None else g.data_ptr(), b.offsets[i], p.numel()]
            mx = max(mx, p.numel())
        if b.flat.is_cuda:
            table = ops.table_for(b.table, 0, tuple(rows), rows, b.flat.device)
            ops.pack_scaled_multi(table, len(b.params), mx, b.flat, scale)
        elif DataParallel.HOST_STAGER is not None:   # installed by the gloo plumbing tests only (tests/test_dist_gloo.py)
            DataParallel.HOST_STAGER(b.params, views, scale)
        else:
            raise SsecgError("DataParallel: host tensors - the HIP library addresses device memory only (no CPU path)")
        for i, p in enumerate(b.params):
            p.grad = views[i]      # also where this rank produced none: the ranks' optimisers must see the same gradients
        if SF.COLLECTIVE_LOG is not None:   # one log with the SyncBN collectives: the tests compare the ranks' issue order
            SF.COLLECTIVE_LOG.append(("grad_bucket", b.flat.numel(), str(b.flat.dtype)))
        b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.process_group, async_op=True)
        b.launched = True

    def _finish(self):
        """End of the backward pass: buckets some parameter of which received no gradient are reduced now (zeros in the missing
        slots - the other ranks may have used the parameter), every collective is waited for, the counters are re-armed."""
        try:
            late = self._buckets[self._next:]
            if late and SF.COLLECTIVES_ISSUED[0] != self._bn_seen:
                missing = [n for n, p in self.module.named_parameters() if p in self._slot and p.grad is None][:4]
                raise SsecgError(
                    "DataParallel: a parameter received no gradient on this rank (" + ", ".join(missing) + " ...) while SyncBatchNorm "
                    "all-reduces were issued during the same backward pass: the other ranks have already issued this bucket between "
                    "their BatchNorm collectives, so the collective order on the process group differs between ranks.  Every "
                    "trainable parameter must take part in every backward when ddp.sync_bn is on (the reference's "
                    "DistributedDataParallel raises for any unused parameter).  The process group now holds unmatched "
                    "collectives: this process must exit (no recovery).")
            for b in late:
                self._launch(b)
            for b in self._buckets:
                if b.work is not None:
                    b.work.wait()
        finally:
            for b in self._buckets:
                b.pending, b.work, b.launched = len(b.params), None, False
            self._armed, self._next = False, 0


def unwrap(m):
    """The module inside a data-parallel wrapper (this one or torch's), or ``m`` itself."""
    return m.module if isinstance(m, (DataParallel, torch.nn.parallel.DistributedDataParallel)) else m
