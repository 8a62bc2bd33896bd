"""Multi-tensor AdamW and EMA on the HIP kernels.

``FusedAdamW`` is a ``torch.optim.Optimizer`` whose ``state_dict`` has the layout
of ``torch.optim.AdamW`` (per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq``) so
checkpoints written by the reference (``src/utils/misc.py:281-302``) resume here and
vice versa.  One kernel launch updates every tensor of a param group: the host
only uploads a small pointer table (rebuilt when a ``.grad`` tensor moved).

``ema_update`` restates ``src/algorithms/mean_teacher.py:138-149``: every parameter
AND every buffer; an int64 ``num_batches_tracked`` on the teacher becomes float32
at the first update exactly as the reference's ``buffer_k.data = ...`` does.
"""
from __future__ import annotations

import torch

from . import ops
from .lib import SsecgError


def _drop_grads_if_capture_fails(param_groups):
    """While a step is captured into a HIP graph (ssecg/graph.py) the backward only RECORDS launches: every ``.grad`` points at
    capture-pool memory no kernel has written.  A capture that fails re-runs the step eagerly, and AccumulateGrad would add
    into that memory - the abort path drops the gradients instead (also registered by the loss scaler, which runs
    the backward itself)."""
    if ops.STEP_SCALARS is not None:
        plist = [p for g in param_groups for p in g["params"]]
        ops.STEP_SCALARS.aborts.append(lambda plist=plist: [setattr(p, "grad", None) for p in plist])


class _FusedOptimizer(torch.optim.Optimizer):
    """Shared plumbing of the fused optimisers: the gradient pointer table (for the global-norm / GradScaler kernel), the
    device-side inf-skip flag, and the lazy reconciliation of per-parameter ``step`` counters after skipped updates."""
    supports_found_inf = True

    def _init_fused(self):
        self._tables = {}
        self._grad_table = {}
        self._skipped = None          # device floats, one per param group: launches skipped on the device (THIS optimiser's)
        self._skips_applied = [0] * len(self.param_groups)

    def _skipped_counter(self, device):
        if self._skipped is None or self._skipped.device != device or self._skipped.numel() != len(self.param_groups):
            self._skipped = torch.zeros(len(self.param_groups), dtype=torch.float32, device=device)
            self._skips_applied = [0] * len(self.param_groups)
        return self._skipped

    def _plist(self, group):
        plist = [p for p in group["params"] if p.grad is not None]
        for p in plist:
            if not p.is_cuda or p.dtype != torch.float32:
                raise SsecgError(f"{type(self).__name__}: parameters must be fp32 HIP tensors (no CPU fallback)")
            if not p.is_contiguous() or not p.grad.is_contiguous():
                raise SsecgError(f"{type(self).__name__}: non-contiguous parameter or gradient")
        return plist

    def grad_table(self):
        """-> (device table of rows {grad*, numel}, ntensors, max_numel, total_numel) over every parameter with a gradient."""
        ptrs, mx, tot, dev = [], 0, 0, None
        for group in self.param_groups:
            for p in self._plist(group):
                ptrs += [p.grad.data_ptr(), p.numel()]
                mx = max(mx, p.numel()); tot += p.numel(); dev = p.device
        if not ptrs:
            return None
        return ops.table_for(self._grad_table, 0, tuple(ptrs), ptrs, dev), len(ptrs) // 2, mx, tot

    @torch.no_grad()
    def grad_norm(self, scaler_state=None, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000, max_norm=None):
        """Global L2 gradient norm [+ GradScaler.update() + clip_grad_norm_] on the device -> tensor [norm, found_inf]."""
        gt = self.grad_table()
        if gt is None:
            return None
        table, n, mx, tot = gt
        out = ops.grad_norm_multi(table, n, 2, 0, 1, mx, scaler_state, growth_factor, backoff_factor, growth_interval, tot)
        if max_norm is not None:
            ops.grad_clip_multi(table, n, 2, 0, 1, mx, out, max_norm)
        return out

    def reconcile_skips(self):
        """Launches skipped on the device (non-finite gradients) did not happen for torch's per-parameter ``step`` either
        (GradScaler.step does not call optimizer.step): subtract them.  The counter is this optimiser's own (one scaler may
        serve two optimisers, as in CPS).  Reads one device float -> only called at checkpoint time; between checkpoints
        the kernel itself subtracts the counter for its bias corrections."""
        if self._skipped is None:
            return
        skipped = [int(v) for v in self._skipped.tolist()]
        for gi, group in enumerate(self.param_groups):
            d = skipped[gi] - self._skips_applied[gi]
            if d > 0:
                for p in group["params"]:
                    st = self.state.get(p)
                    if st and "step" in st:
                        st["step"] -= d
                self._skips_applied[gi] = skipped[gi]

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        # the loaded ``step`` counts real steps only.  The counter tensor is zeroed IN PLACE: a captured step (ssecg/graph.py)
        # holds its address
        if self._skipped is not None and self._skipped.numel() == len(self.param_groups):
            self._skipped.zero_()
        else:
            self._skipped = None
        self._skips_applied = [0] * len(self.param_groups)
        self._tables = {}
        self._grad_table = {}

    def state_dict(self):
        self.reconcile_skips()
        return super().state_dict()


class FusedAdamW(_FusedOptimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        if lr < 0.0 or eps < 0.0 or not (0.0 <= betas[0] < 1.0) or not (0.0 <= betas[1] < 1.0) or weight_decay < 0.0:
            raise ValueError("invalid AdamW hyper-parameter")
        defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False,
                        maximize=False, foreach=None, capturable=False, differentiable=False, fused=None)
        super().__init__(params, defaults)
        self._init_fused()

    def _table(self, gi, plist):
        """Device pointer table for group gi; re-uploaded only when a pointer changed."""
        ptrs = []
        for p in plist:
            st = self.state[p]
            ptrs += [p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel()]
        return ops.table_for(self._tables, gi, tuple(ptrs), ptrs, plist[0].device), max(p.numel() for p in plist)

    @torch.no_grad()
    def step(self, closure=None, found_inf=None):
        """``found_inf``: device float tensor; non-zero -> this update is skipped inside the kernel (no host sync)."""
        loss = None
        _drop_grads_if_capture_fails(self.param_groups)
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            plist = self._plist(group)
            if not plist:
                continue
            for p in plist:
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            table, max_numel = self._table(gi, plist)
            skipped = self._skipped_counter(plist[0].device) if found_inf is not None else self._skipped
            if skipped is not None and len(self._skips_applied) != len(self.param_groups):
                skipped = self._skipped_counter(plist[0].device)     # a param group was added

            def host_step(gi=gi, group=group, plist=plist, counted=skipped is not None):
                """The host half of one step of this group: advance the per-parameter ``step`` -> the launch's scalars.  Also
                the refresh hook of a step captured in a HIP graph (ssecg/graph.py): a replay runs this and nothing else."""
                steps = {int(self.state[p]["step"].item()) for p in plist}  # CPU scalars: no device sync
                if len(steps) != 1:
                    raise SsecgError("FusedAdamW: parameters of one group must share a step count")
                t = steps.pop() + 1
                for p in plist:
                    self.state[p]["step"] += 1
                b1, b2 = group["betas"]
                # ``t`` counts launches; the ones the device skipped are subtracted in the kernel (bias corrections) and, at
                # checkpoint time, from the per-parameter ``step`` (reconcile_skips)
                return (float(group["lr"]), float(b1), float(b2), float(group["weight_decay"]),
                        t + (self._skips_applied[gi] if counted else 0))

            lr, b1, b2, wd, t = host_step()
            if ops.STEP_SCALARS is not None:   # a capture that fails must leave the host bookkeeping where it found it
                ops.STEP_SCALARS.aborts.append(lambda plist=plist: [self.state[p]["step"].sub_(1) for p in plist])
            ops.adamw_multi(table, len(plist), max_numel, lr, b1, b2, float(group["eps"]), wd, t,
                            total_numel=sum(p.numel() for p in plist), skip_flag=found_inf,
                            skipped_count=(skipped[gi:gi + 1] if skipped is not None else None), refresh=host_step)
        return loss


class FusedSGD(_FusedOptimizer):
    """``torch.optim.SGD(lr, momentum, weight_decay)`` as ``src/utils/optimizer.py:15-26`` builds it (dampening 0, no
    nesterov), one launch per param group; ``state_dict`` has torch's layout (``momentum_buffer`` per parameter)."""

    def __init__(self, params, lr=1e-3, momentum=0.0, weight_decay=0.0):
        if lr < 0.0 or momentum < 0.0 or weight_decay < 0.0:
            raise ValueError("invalid SGD hyper-parameter")
        defaults = dict(lr=lr, momentum=momentum, dampening=0, weight_decay=weight_decay, nesterov=False, maximize=False,
                        foreach=None, differentiable=False, fused=None)
        super().__init__(params, defaults)
        self._init_fused()

    @torch.no_grad()
    def step(self, closure=None, found_inf=None):
        loss = None
        _drop_grads_if_capture_fails(self.param_groups)
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            plist = self._plist(group)
            if not plist:
                continue
            mom = float(group["momentum"])
            first = False
            if mom != 0.0:
                have = [self.state[p].get("momentum_buffer") is not None for p in plist]
                if any(have) != all(have):
                    raise SsecgError("FusedSGD: parameters of one group must all have (or all lack) a momentum buffer")
                first = not all(have)
                if first:
                    for p in plist:
                        self.state[p]["momentum_buffer"] = torch.empty_like(p, memory_format=torch.preserve_format)
            ptrs = []
            for p in plist:
                buf = self.state[p].get("momentum_buffer") if mom != 0.0 else None
                ptrs += [p.data_ptr(), p.grad.data_ptr(), buf.data_ptr() if buf is not None else 0, p.numel()]
            table, max_numel = ops.table_for(self._tables, gi, tuple(ptrs), ptrs, plist[0].device), max(p.numel() for p in plist)
            if first and found_inf is not None:
                # a skipped FIRST step must leave the buffers "absent": rare enough for one host read
                if float(found_inf.reshape(-1)[0].item()) != 0.0:
                    for p in plist:
                        self.state[p].pop("momentum_buffer", None)
                    self._tables.pop(gi, None)
                    continue
            ops.sgd_multi(table, len(plist), max_numel, float(group["lr"]), mom, float(group["weight_decay"]), first,
                          total_numel=sum(p.numel() for p in plist), skip_flag=found_inf,
                          refresh=lambda group=group: float(group["lr"]))
        return loss


class EmaUpdater:
    """teacher <- decay*teacher + (1-decay)*student over parameters and buffers, one launch."""

    def __init__(self):
        self._cache = {}

    @torch.no_grad()
    def __call__(self, student: torch.nn.Module, teacher: torch.nn.Module, decay: float):
        pairs = list(zip(student.parameters(), teacher.parameters())) + list(zip(student.buffers(), teacher.buffers()))
        # the reference rebinds ``.data`` every step; integer teacher buffers turn float32 on the first update (Q5)
        for s, t in pairs:
            if not t.dtype.is_floating_point:
                t.data = t.data.to(torch.float32)
            elif s.data_ptr() == t.data_ptr():
                t.data = t.data.clone()  # teacher aliased the student at construction (Q4): un-alias like `.data =` does
        ptrs = []
        for s, t in pairs:
            if not (s.is_cuda and t.is_cuda):
                raise SsecgError("ema_update: tensors must live on the HIP device (no CPU fallback)")
            if t.dtype != torch.float32 or s.dtype not in (torch.float32, torch.int64):
                raise SsecgError(f"ema_update: unsupported dtypes {s.dtype} -> {t.dtype}")
            ptrs += [t.data_ptr(), s.data_ptr(), t.numel(), 1 if s.dtype == torch.int64 else 0]
        table = ops.table_for(self._cache, 0, tuple(ptrs), ptrs, pairs[0][1].device)
        ops.ema_multi(table, len(pairs), max(t.numel() for _, t in pairs), float(decay))
