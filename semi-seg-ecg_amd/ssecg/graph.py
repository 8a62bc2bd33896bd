"""A whole training step as ONE HIP graph (``train.hip_graph: true`` / ``bench.py --graph``; ``auto``, the default, switches it on
for runs of at most 128 windows per loader per GPU, on one GPU or over RCCL - algorithms/base.py::resolve_hip_graph).

A FixMatch step is ~250 kernel launches.  At 512 windows per GPU the device needs 22 ms for them and the 7 ms the host
spends enqueueing hide behind it; at the reference's small-batch configurations (BASELINE configs #2 / #4: 256 windows per
GPU, `use_amp: true`) the device needs under 6 ms and the step is bound by the host.  ``StepGraph`` captures the step -
teacher pass, student pass, losses, backward, gradient norm / GradScaler update, optimiser - once, after two eager warm-up
steps, and replays it with one ``hipGraphLaunch`` per step.

What changes from step to step on the host and how it reaches the replayed kernels:

* the batch: copied into the graph's static input tensors (device to device) before the replay;
* the learning rate and AdamW's step count (bias corrections): ``ssecg_adamw_multi`` takes ``coef_dev`` - five doubles formed
  by ``ssecg_adamw_coefficients`` with the arithmetic of the by-value launch - from a small device block (``StepScalars``);
* dropout seeds: drawn from torch's CPU generator exactly as the eager head does (same number of draws, same order) and read
  by ``ssecg_dropout_fwd`` through ``seed_dev``.

Ownership: every pointer table a captured launch reads comes out of the capturing graph's OWN block
(``ops.table_for``), and the graph holds strong references (``StepScalars.keep``) to every tensor outside the step's closure
whose address a captured launch uses - the operands and weights of EVERY model registered in the process-global operand caches,
because the captured refresh launches cover them all.  A model freed while a graph that saw it keeps replaying therefore
cannot have its blocks recycled under the replay.  A capture that fails (an op that cannot be captured, more per-step scalars
than the block holds, per-launch timing switched on) rolls the host bookkeeping back (per-parameter ``step``, torch's CPU
generator), logs once, and the StepGraph runs its step eagerly from then on.

Each consumer registers a ``refresh`` callable when it asks for its slot during capture; before every replay the callables
run in capture order (they also do the consumer's host-side bookkeeping, e.g. the optimiser's per-parameter ``step``), the
block is uploaded with one 4 KB copy on the step's stream, and the graph is launched.  A replayed step is therefore
bit-identical to the eager step (``tests/test_graph_gpu.py``).  Not captured: data loading / augmentation, the learning-rate
schedule, metric logging (its packed all-reduce included).  Under ``torch.distributed`` over RCCL (round 6) the step is captured
WITH its collectives: the SyncBatchNorm all-reduces and the gradient buckets of ``ssecg.parallel.DataParallel`` are
ProcessGroupNCCL launches on the backend's stream, their ``work.wait()`` a stream dependency - graph nodes and edges; the
reducer's host bookkeeping (bucket countdown, ``.grad`` re-pointed into the flat buffers) runs once, during the capture, and a
replay issues the same collectives in the same order on every rank.  Not supported (the eager path is taken): gloo groups,
torch's own DistributedDataParallel reducer, gradient accumulation, batches whose shapes differ from the captured ones.
"""
from __future__ import annotations

import ctypes
import time

import torch
import torch.distributed as dist

from . import ops
from .lib import SsecgError, check, lib

#: Seconds a capture under RCCL waits, after a device synchronize, before it begins (round 6).  ProcessGroupNCCL's watchdog thread
#: wakes every 100 ms, polls the end event of every collective still on its list and only then drops the completed ones.  The eager
#: warm-up step in front of the capture leaves ~40 works there; if the watchdog's next poll falls into the first milliseconds of the
#: capture - while RCCL's stream is being pulled into it - hipEventQuery fails with hipErrorCapturedEvent ("operation not permitted on an
#: event last recorded in a capturing stream"), the exception ends the watchdog thread and with it the PROCESS (SIGABRT; nothing
#: Python can catch).  Measured with tools/probes/rccl_step_graph_repeat.py on the one-rank RCCL step: 3 aborts in 60 captures
#: without the wait, all inside the ~10 ms capture; 0 in 60 with it (and 0 when a poll falls LATER into a capture held open for
#: 250 ms: works issued during the capture are not put on the list).  Measured with 0.25 s; 0.35 s = 3.5 watchdog periods for margin
#: on a loaded host: its list is empty when the capture begins.  Once per captured step function, i.e. once per training stage.
NCCL_WATCHDOG_DRAIN_S = 0.35


def _nccl_group_active() -> bool:
    """True if the default process group runs (also) on RCCL - "nccl" or a per-device map such as "cpu:gloo,cuda:nccl"."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    try:
        return "nccl" in str(dist.get_backend()).lower() or "nccl" in str(dist.get_backend_config()).lower()
    except Exception:  # noqa: BLE001 - an API difference must not cost the drain
        return True


_WORDS = 512  # 8-byte words per block (4 KB): an AdamW group takes 5, a seed or a learning rate 1 - ~100 param groups
_TABLE_WORDS = 1 << 15   # int64 words for the pointer tables built during the capture (a ResNet18 step needs ~2 k)


class StepScalars:
    """The device block of per-step host scalars and the recipe to refill it."""

    RING = 8   # pinned host blocks: the device runs several steps behind the host, an upload must not be overwritten in flight

    def __init__(self, device):
        self.hosts = [torch.zeros(_WORDS, dtype=torch.float64).pin_memory() for _ in range(self.RING)]
        self.events = [None] * self.RING
        self.slot = 0
        self.host = self.hosts[0]
        self.host_i = self.host.view(torch.int64)
        self.dev = torch.zeros(_WORDS, dtype=torch.float64, device=device)
        self.used = 0
        self.refreshers = []          # (kind, offset, callable) in capture order
        self.keep = []                # tensors whose addresses the captured launches use (ops.keep_for_graph)
        self.aborts = []              # undo callables of host bookkeeping done during the capture (run if it fails)
        # pointer tables built while the step is captured (optimiser, EMA, weight operands: their tensors move into the graph's
        # memory pool): slices of ONE pinned block and ONE device block allocated before the capture starts; append-only, so the
        # captured copies find their sources unchanged at every replay
        self.tab_host = torch.zeros(_TABLE_WORDS, dtype=torch.int64).pin_memory()
        self.tab_dev = torch.zeros(_TABLE_WORDS, dtype=torch.int64, device=device)
        self.tab_used = 0

    def _take(self, n):
        if self.used + n > _WORDS:
            raise SsecgError("StepScalars: more per-step scalars than the block holds")
        k = self.used
        self.used += n
        return k

    def _put_adamw(self, k, lr, beta1, beta2, weight_decay, step):
        out = (ctypes.c_double * 5)()
        check(lib().ssecg_adamw_coefficients(float(lr), float(beta1), float(beta2), float(weight_decay), int(step),
                                             ctypes.cast(out, ctypes.c_void_p)), "ssecg_adamw_coefficients")
        for j in range(5):
            self.host[k + j] = out[j]

    def take_table(self, rows):
        n = len(rows)
        k = (self.tab_used + 1) & ~1          # 16-byte aligned slices
        if k + n > _TABLE_WORDS:
            raise SsecgError("StepScalars: pointer tables of the captured step exceed the pre-allocated block")
        self.tab_host[k:k + n] = torch.tensor(rows, dtype=torch.int64)
        dev = self.tab_dev[k:k + n]
        dev.copy_(self.tab_host[k:k + n], non_blocking=True)
        self.tab_used = k + n
        return dev

    # ---- called by ssecg.ops while the step is being captured -> device address of the slot
    def adamw_slot(self, lr, beta1, beta2, weight_decay, step, refresh=None):
        if refresh is None:
            raise SsecgError("a captured AdamW launch needs its optimiser's refresh hook")
        k = self._take(5)
        self._put_adamw(k, lr, beta1, beta2, weight_decay, step)
        self.refreshers.append(("adamw", k, refresh))
        return self.dev.data_ptr() + 8 * k

    def lr_slot(self, lr, refresh=None):
        if refresh is None:
            raise SsecgError("a captured SGD launch needs its optimiser's refresh hook")
        k = self._take(1)
        self.host[k] = float(lr)
        self.refreshers.append(("lr", k, refresh))
        return self.dev.data_ptr() + 8 * k

    def seed_slot(self, seed, refresh=None):
        k = self._take(1)
        self.host_i[k] = _as_i64(seed)
        self.refreshers.append(("seed", k, refresh if refresh is not None else ops.draw_seed))
        return self.dev.data_ptr() + 8 * k

    # ---- before a replay
    def refresh(self):
        self.slot = (self.slot + 1) % self.RING
        if self.events[self.slot] is not None:
            self.events[self.slot].synchronize()      # its last upload (RING steps ago) has long completed; never blocks in practice
        self.host = self.hosts[self.slot]
        self.host_i = self.host.view(torch.int64)
        for kind, k, fn in self.refreshers:
            v = fn()
            if kind == "adamw":
                self._put_adamw(k, *v)
            elif kind == "lr":
                self.host[k] = float(v)
            else:
                self.host_i[k] = _as_i64(v)

    def upload(self):
        self.dev.copy_(self.host, non_blocking=True)
        if self.events[self.slot] is None:
            self.events[self.slot] = torch.cuda.Event()
        self.events[self.slot].record()


def _as_i64(seed: int) -> int:
    seed = int(seed) & (2 ** 64 - 1)
    return seed - 2 ** 64 if seed >= 2 ** 63 else seed


class StepGraph:
    """``step_fn(*tensors) -> tuple of tensors`` (forward, backward, optimiser, zero_grad) captured after ``warmup`` eager
    calls and replayed from then on.  Calls whose input shapes differ from the captured ones run eagerly."""

    def __init__(self, step_fn, warmup: int = 2):
        self.step_fn = step_fn
        self.warmup = warmup
        self.calls = 0
        self.graph = None
        self.disabled = False         # a failed capture: eager from then on
        self.scalars = None
        self.static_in = None
        self.static_out = None
        self.replays = 0

    def _sig(self, inputs):
        return tuple((tuple(t.shape), t.dtype, t.device) for t in inputs)

    def __call__(self, *inputs):
        self.calls += 1
        if self.disabled:
            return self.step_fn(*inputs)
        if self.graph is None:
            if self.calls <= self.warmup:
                return self.step_fn(*inputs)
            if not self._capture(inputs):
                return self.step_fn(*inputs)   # capture failed and was rolled back: this step (and every later one) runs eagerly
        elif self._sig(inputs) != self.sig:
            return self.step_fn(*inputs)       # e.g. a short last batch
        else:
            for s, t in zip(self.static_in, inputs):
                s.copy_(t, non_blocking=True)
            self.scalars.refresh()
        self.scalars.upload()
        self.graph.replay()
        self.replays += 1
        return self.static_out

    def _capture(self, inputs):
        """-> True if the step is captured (the caller replays it), False if the capture failed and was rolled back."""
        dev = inputs[0].device
        self.sig = self._sig(inputs)
        self.static_in = [torch.empty_like(t) for t in inputs]
        for s, t in zip(self.static_in, inputs):
            s.copy_(t)
        self.scalars = StepScalars(dev)
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize(dev)
        if _nccl_group_active():
            time.sleep(NCCL_WATCHDOG_DRAIN_S)   # the watchdog drops the eager steps' completed works (see NCCL_WATCHDOG_DRAIN_S)
        rng = torch.get_rng_state()          # dropout seeds are drawn from torch's CPU generator during the capture
        ops.STEP_SCALARS = self.scalars
        try:
            if ops.PROFILE is not None:
                raise SsecgError("StepGraph: per-launch timing (ops.PROFILE) cannot be captured")
            # relaxed: the optimiser re-uploads its pointer tables (pinned host allocation + copy) when the gradients move
            # into the graph's memory pool
            with torch.cuda.graph(g, capture_error_mode="relaxed"):
                out = self.step_fn(*self.static_in)
        except Exception as e:  # noqa: BLE001 - any failure: the captured kernels never ran, undo the host side and go eager
            ops.STEP_SCALARS = None
            for undo in reversed(self.scalars.aborts):
                undo()
            torch.set_rng_state(rng)
            self.disabled, self.scalars, self.static_in, self.static_out = True, None, None, None
            print(f"ssecg.graph: capturing the step failed ({type(e).__name__}: {e}); running it eagerly from now on", flush=True)
            torch.cuda.synchronize(dev)
            return False
        finally:
            ops.STEP_SCALARS = None
        self.static_out = out
        self.graph = g
        return True

    def release(self):
        """Drop the graph, its memory pool and every tensor it kept alive (a training stage that ends: ST++)."""
        self.graph, self.scalars, self.static_in, self.static_out = None, None, None, None
        self.disabled = True
