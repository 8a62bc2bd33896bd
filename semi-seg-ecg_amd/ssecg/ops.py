"""Tensor-level wrappers over the C ABI (no autograd here).

PyTorch-ROCm is used for device memory and streams only: every function
validates its operands, allocates outputs with ``torch.empty`` and enqueues the
HIP kernel on the caller's current stream.  CPU tensors are rejected - the
product path has no CPU implementation.
"""
from __future__ import annotations

import weakref

import torch

from . import config
from .lib import SsecgError, check, lib, trace


#: while a training step is being CAPTURED into a HIP graph (ssecg/graph.py): the per-step host scalars (learning rate, AdamW
#: step count, dropout seeds) are read by the kernels from this device block instead of being frozen into the launches
STEP_SCALARS = None

#: when a list, every conv launch appends (kernel name, algorithmic FLOPs, start event, end event) - bench.py's
#: live per-kernel timing on the launch stream; None (default) = no instrumentation.
PROFILE = None


def _igemm_symbol(m: int, csrc: int, k: int, mode: int, stride: int = 1) -> str:
    """Kernel symbol a conv launch resolves to (mirrors fast_ok / pick_cfg / dgrad_phased in csrc/conv.hip), so that
    bench.py's per-kernel timing lines up with the names rocprofv3 reports."""
    fast = m > 32 and csrc % 16 == 0 and k in (1, 3)
    if fast and not (mode == 1 and stride == 2):
        t = "256, 128, 4, 2" if m > 128 else ("128, 128, 2, 4" if m > 64 else "64, 512, 1, 8")
        return f"conv_igemm_fast_kernel<{t}, {k}>"
    if fast:  # stride-2 data gradient: two phase launches (k3 -> taps {1} and {0,2}; k1 -> one phase after a memset)
        t = "256, 128, 4, 2" if m > 128 else ("128, 128, 2, 4" if m > 64 else "64, 512, 1, 8")
        return f"conv_igemm_fast_kernel<{t}, 1>" + (f" + <{t}, 2> (stride-2 dgrad phases)" if k == 3 else " (stride-2 dgrad)")
    t = "128, 128, 2, 2" if m > 64 else ("64, 256, 1, 4" if m > 32 else "32, 256, 1, 4")
    return f"conv_igemm_kernel<{t}, {k}, {mode}>"


class _Timed:
    def __init__(self, name, flops, nbytes=0.0):
        """flops / nbytes: ALGORITHMIC work of the launch: direct-conv FLOPs, operands + result bytes."""
        self.on = PROFILE is not None
        if self.on:
            self.name, self.flops, self.nbytes = name, flops, nbytes
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def __enter__(self):
        if self.on:
            self.e0.record()
        return self

    def __exit__(self, *a):
        if self.on:
            self.e1.record()
            PROFILE.append((self.name, self.flops, self.e0, self.e1, self.nbytes))
        return False


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream() -> int:
    """hipStream_t of torch's current stream on the current device.  The raw getter is ~20x cheaper than building a
    torch.cuda.Stream object per launch (280 launches per step: 0.9 ms of host time at small batch)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _req(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise SsecgError(f"{name}: expected a HIP device tensor (the MI355X hot path has no CPU fallback)")
    if t.dtype != dtype:
        raise SsecgError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _p(t):
    return None if t is None else t.data_ptr()


def upload_table(rows, device):
    """int64 pointer / shape table -> device tensor (one small asynchronous copy from pinned memory).  While a step is being
    captured into a HIP graph the table comes out of the graph's own pre-allocated block instead: pinned allocations and frees
    inside a capture trip the host allocator's event bookkeeping (hipErrorCapturedEvent), and the captured copy must find its
    source unchanged at every replay."""
    if STEP_SCALARS is not None:
        return STEP_SCALARS.take_table(rows)
    return torch.tensor(rows, dtype=torch.int64).pin_memory().to(device, non_blocking=True)


def table_for(cache, slot, key, rows, device):
    """Device pointer table for ``rows``, cached as ``cache[slot] = (key, table)`` and re-uploaded only when ``key`` changed.
    While a step is being captured into a HIP graph (``STEP_SCALARS`` set) the table ALWAYS comes fresh out of the capturing
    graph's own block and the cache is left alone: a cached table belongs to the cache only - an eager call that
    changes the key later would free it under the graph that baked its address in; and a second graph with the same key must
    not read the first graph's block."""
    if STEP_SCALARS is not None:
        return STEP_SCALARS.take_table(rows)
    ent = cache.get(slot)
    if ent is None or ent[0] != key:
        ent = (key, upload_table(rows, device))
        cache[slot] = ent
    return ent[1]


def keep_for_graph(*tensors):
    """While a step is being captured: the graph keeps a strong reference to every tensor whose ADDRESS one of its captured
    launches uses but which the step's own closure does not own (transformed-weight operands, folded BatchNorm coefficients,
    and the weights of every model registered in the process-global operand caches - the refresh launches cover them all).  A
    model that is freed while a graph that saw it keeps replaying would otherwise hand its blocks to the allocator and the
    replay would write into whatever lives there next."""
    if STEP_SCALARS is not None:
        STEP_SCALARS.keep.extend(t for t in tensors if t is not None)


class BatchPair(tuple):
    """Two batches that a model consumes as ONE batch along dim 0 - ``torch.cat((a, b))`` without the copy: the stem convolution
    (and its weight gradient) read the two tensors through two source pointers.  ``cat()`` materialises it for everything else."""

    def __new__(cls, a, b):
        if a.dim() != 3 or a.shape[1:] != b.shape[1:] or a.dtype != b.dtype or a.device != b.device:
            raise SsecgError("BatchPair: the two batches must agree in (C, L), dtype and device")
        return super().__new__(cls, (a, b))

    def size(self, dim=None):
        shp = torch.Size((self[0].shape[0] + self[1].shape[0],) + tuple(self[0].shape[1:]))
        return shp if dim is None else shp[dim]

    @property
    def shape(self):
        return self.size()

    @property
    def device(self):
        return self[0].device

    def cat(self):
        return torch.cat((self[0], self[1]))


def batch_pair(a, b):
    """What the plugins pass to the student model instead of ``torch.cat((a, b))`` (src/algorithms/fixmatch.py:98-100)."""
    if STEM_PAIR and a.is_cuda and a.dim() == 3 and b.shape[0] > 0 and a.shape[0] > 0:
        return BatchPair(a, b)
    return torch.cat((a, b))


def conv_out_len(lin: int, k: int, stride: int, pad: int, dil: int = 1) -> int:
    return (lin + 2 * pad - dil * (k - 1) - 1) // stride + 1


# ----------------------------------------------------------------------------- conv
#: 3-tap stride-1 convolutions (forward and data gradient) run in Winograd F(2,3) form: 2/3 of the multiplications on the
#: same fp32 matrix pipe.  SSECG_WINOGRAD=0 keeps every conv on the direct implicit-GEMM kernels.
WINOGRAD = config.switch("SSECG_WINOGRAD", True, "3-tap stride-1 convolutions in Winograd form; 0 = direct implicit GEMM everywhere", __name__, "WINOGRAD")
# Transformed-weight operands.  Correctness rule: an operand is NEVER reused across a point where the weights could have
# been rewritten without this module seeing it (``param.data`` edits have their own version counter; the reference itself
# rebinds ``.data``, src/algorithms/mean_teacher.py:144; optimisers write through raw pointers).  Therefore
#   * a MODEL forward (EncoderDecoder / ResNet / FCNHead .forward) calls ``begin_forward()``: every operand made before
#     it is stale; the first Winograd conv after it re-transforms ALL registered weights in ONE launch (0.02-0.05 ms) and
#     the rest of that forward and its backward (``w_cached=True`` from ssecg.functional) reuse them;
#   * a STANDALONE op call (``w_cached=False``, the default of conv1d_fwd / conv1d_dgrad) re-transforms its weight every
#     time (one small launch) - nothing is trusted.
# There is no obligation on callers (the round-1 ``weights_changed()`` duty is gone; the function remains as a no-cost hint).
#: Winograd variant for those convolutions: 4 = F(4,3) (six multiplications per four outputs: half the direct form's MFMA
#: work), 2 = F(2,3) (eight).  SSECG_WINO_F selects.
WINO_F = config.switch("SSECG_WINO_F", 4, "Winograd variant of forward / data gradient: 4 = F(4,3), 2 = F(2,3)", __name__, "WINO_F", (2, 4))
#: Winograd variant of their weight gradient (both channel counts multiples of 128): 4 = the transpose of F(4,3) (six
#: multiplications per four positions; round 5: 16 waves with the six planes split between two wave groups, 16-byte staging - the
#: 8-wave form of round 2, tools/experiments/r04_wino4_wgrad.patch, measured slower than F(2,3)), 2 = the transpose of F(2,3).
WINO_WGRAD_F = config.switch("SSECG_WINO_WGRAD_F", 4, "Winograd variant of the weight gradient: transpose of F(4,3) or F(2,3)", __name__, "WINO_WGRAD_F", (2, 4))
#: 2 = only the 64-channel layer back on the 16-wave F(2,3) kernel (the A/B switch of the round-4 change)
WINO_F64 = config.switch("SSECG_WINO_F64", 4, "Winograd variant of the 64-channel layer", __name__, "WINO_F64", (2, 4))
#: K split of small convolution launches (SSECG_KSPLIT, default on; 0 = never).  Batches of 16-64 windows - the reference's shipped
#: batch_size is 16 (configs/base/resnet18/fixmatch.yaml:86) - leave most CUs without a tile (layer4 at N = 32: 32 tiles for 256 CUs,
#: each contracting all 512 channels one 16-channel stage after the other); with the split up to 8 workgroup columns share the
#: channels and one pass sums their partials, applies the epilogue and emits the BatchNorm sums.  The library decides per launch
#: (fewer tiles than workgroup slots: nothing at 512 windows qualifies) - round 5: F(4,3) launches with statistics / fused input
#: BN and the implicit-GEMM launches (stride-2, 1x1, phase data gradients) are split too, and it is on by default.  It changes
#: the summation order with the batch size (tests/test_fullsize_gpu.py compares a window's logits across batch sizes at the
#: kernel bar where a split is involved, bit for bit where none is).
KSPLIT = config.switch("SSECG_KSPLIT", True, "K split of small convolution launches (training passes; evaluate() never splits)", __name__, "KSPLIT")
#: the student batch (labelled, unlabelled) of the semi-supervised plugins travels as a ``BatchPair`` and the stem reads the two
#: tensors where they lie (ssecg_stem_fwd2 / _wgrad2); 0 = concatenate first, as the reference does (same values, bit for bit)
STEM_PAIR = config.switch("SSECG_STEM_PAIR", True, "the student batch travels as a BatchPair (no torch.cat copy)", __name__, "STEM_PAIR")
#: use_amp: the stem's BN + ReLU + MaxPool pass writes the blocked bf16 layout itself (round 4); 0 = fp32 pooled tensor + a
#: separate layout pass (bit-identical values)
AMP_STEM_BLOCKED = config.switch("SSECG_AMP_STEM_BLOCKED", True, "use_amp: the stem pooling pass writes blocked bf16 itself", __name__, "AMP_STEM_BLOCKED")
#: use_amp: the stem convolution on 16-bit OPERANDS with a 16-bit stored output, as autocast runs it (round 5: x, w rounded to bf16
#: while staged, output rounded before the BatchNorm sums; the weight gradient rounds x and dc).  0 = the fp32 stem of rounds 2-4
#: (more precise than the reference under autocast: its pooled output differs from the reference's in 35 % of the elements by one ulp)
AMP_STEM_LP = config.switch("SSECG_AMP_STEM_LP", True, "use_amp: the stem convolution on 16-bit operands, as under autocast", __name__, "AMP_STEM_LP")
#: ... and its (bf16-valued) output c and that output's gradient dc STORED as bf16 (planar): half the bytes of the five passes over the
#: two largest tensors of the stem, identical results bit for bit; 0 = fp32 containers
AMP_STEM_C16 = config.switch("SSECG_AMP_STEM_C16", True, "use_amp: the stem conv output and its gradient stored as bf16", __name__, "AMP_STEM_C16")
#: dedicated kernels for the stem convolution (C -> 64, k 7, stride 2, pad 3); SSECG_STEM=0 routes it through the generic
#: implicit GEMM again (kept for A/B and as the second implementation the tests compare)
STEM = config.switch("SSECG_STEM", True, "dedicated stem kernels; 0 = generic implicit GEMM", __name__, "STEM")
_wino_cache = {}
_weights_epoch = [0]
_scope_depth = [0]
WINO_TRANSFORMS = [0]   # number of weight-transform launches (single + multi)


def _wino_variant(cout, cin):
    """F(4,3) wherever its tiles apply: 8-wave 128 x 64-quad tiles on the 128-channel-multiple layers (15-19 % faster than F(2,3)
    on every such shape) and, since round 4, 64 x 128-quad tiles on the 64-channel layer.  (Rounds 2-3 kept layer1 on the 16-wave
    F(2,3) kernel: the 64 x 128 tile then staged six dword loads per (channel, quad) item and ran at 81 TF against 89 TF.  With
    one 16-byte load + two dword loads per item it fits the register file without spills and measures 125 against 100 TF
    (N = 1024, forward; profiles/r04_wino4_layer1.txt).)"""
    if WINO_F != 4 or cout % 64 != 0 or cin % 64 != 0:
        return 2
    if (cout % 128 != 0 or cin % 128 != 0) and WINO_F64 == 2:
        return 2      # A/B switch: the 64-channel layer back on the 16-wave F(2,3) kernel
    return 4


# ---- the pseudo-label pass beside the student forward (small batches) -----------------------------------------------------
# At the reference's shipped batch size (16 windows per loader, configs/base/resnet18/fixmatch.yaml:86) every kernel of the
# step has tiles for a fraction of the 256 CUs, and the pseudo-label pass (eval mode, frozen statistics, no gradient) and the
# student forward are independent until the loss.  ``PassOverlap`` runs the first on a side HIP stream: fork
# (side.wait_stream(main)) -> teacher pass on the side stream | student forward on the main stream -> join
# (main.wait_stream(side)).  What the two passes share is made BEFORE the fork on the main stream: the Winograd / bf16 weight
# operands and the folded BatchNorm coefficients of the eval pass (which the student's statistics launches would otherwise
# race: they update the running statistics the fold reads).  An operand the caches do not know at the fork (first step of a
# model) is made lazily inside the region, between two fences that order BOTH streams around it (_overlap_fence).  Memory: blocks the side stream allocates return to ITS pool and
# are reused by the next step's teacher pass only, which starts behind that step's fork - i.e. behind every main-stream use
# enqueued before it; no record_stream needed.  Same kernels, same order per stream: results are bit-identical
# (tests/test_graph_gpu.py::test_pass_overlap_is_bit_identical).  Inside a HIP-graph capture the fork / join become graph
# edges, which is where it pays: the eager small-batch step is host-bound either way.
#: SSECG_OVERLAP_PASSES: "auto" / "1" (default) = on at every batch size (512 windows 20.1 -> 20.0 ms fp32, 8.98 -> 8.92 bf16 - the fp32
#: pseudo-label pass is matrix-pipe work, the student pass has the HBM-bound kernels; 16 windows 3.06 -> 2.82), also under
#: torch.distributed: the pseudo-label pass issues no collective, and it fills the stream hand-offs of the student pass's SyncBN
#: all-reduces (one rank over RCCL, collectives forced: 20.45 -> 20.30 ms); "0" = never
OVERLAP_PASSES = config.switch("SSECG_OVERLAP_PASSES", "auto", "pseudo-label pass on a side stream: auto / 1 = on, 0 = never", __name__, "OVERLAP_PASSES",
                                ("auto", "1", "0"))
_side_streams = {}
_overlap_active = {}     # device index -> the PassOverlap whose two streams are running there (fork ... join)


def _overlap_fence(device=None):
    """A shared operand is about to be (or has just been) re-made lazily INSIDE an overlapped region - a weight or BatchNorm the
    caches did not know at the fork (the first step of a model): order both streams around it.  (Found the hard way: the pass
    that registers a weight stamps its operand as current, and the other stream would read it before the transform ran.)
    ``device``: the device whose operands are re-made (its overlap slot; default: the current device)."""
    if not _overlap_active:
        return
    idx = device.index if (device is not None and device.index is not None) else torch.cuda.current_device()
    ov = _overlap_active.get(idx)
    if ov is not None:
        ov.main.wait_stream(ov.side)
        ov.side.wait_stream(ov.main)


def wrapper_broadcasts_buffers(m) -> bool:
    """Does calling ``m`` issue a buffer broadcast at every forward?  A data-parallel wrapper - this library's or torch's
    DistributedDataParallel, which has no ``world_size`` attribute - with ``broadcast_buffers`` set, in a process group
    of more than one rank (``ddp.sync_bn: false``: the ranks follow rank 0's running statistics, src/algorithms/fixmatch.py:292-296
    with DDP's defaults).  That collective, and its write into the running statistics the eval-mode fold reads, must not be issued
    from a side stream beside the student pass's own broadcast."""
    if not getattr(m, "broadcast_buffers", False):
        return False
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(getattr(m, "process_group", None)) > 1


class PassOverlap:
    """The pseudo-label pass on a side HIP stream beside the student forward::

        with ops.PassOverlap(n_windows, device, model) as ov:
            with ov.teacher(), torch.no_grad():
                <pseudo-label pass>                 # side stream
            <student forward>                       # main stream
        # leaving the block JOINS: main waits for side

    Exception-safe (round 6): ``__exit__`` always joins and clears the device's active slot - an exception between the fork and
    the join (an out-of-memory student forward, a KeyboardInterrupt) used to leave the slot set, every later step silently
    single-stream and the main stream never ordered behind the side stream's kernels.  One overlap per device at a time (a nested
    one is off); ``join()`` may be called early and is idempotent."""

    def __init__(self, n_windows, device, *models):
        """``models``: what the two passes call (see ``wrapper_broadcasts_buffers``)."""
        on = OVERLAP_PASSES != "0" and device.type == "cuda" and PROFILE is None
        if on and any(wrapper_broadcasts_buffers(m) for m in models):
            on = False
        self.on, self.device, self.side, self.main, self._key = on, device, None, None, None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.join()
        return False

    def teacher(self):
        return _TeacherSide(self)

    def join(self):
        key, self._key = self._key, None
        if key is not None:
            if _overlap_active.get(key) is self:
                del _overlap_active[key]
            self.main.wait_stream(self.side)


class _TeacherSide:
    def __init__(self, ov):
        self.ov, self.ctx = ov, None

    def __enter__(self):
        ov = self.ov
        dev = ov.device
        if ov.on:
            idx = dev.index if dev.index is not None else torch.cuda.current_device()
            if idx in _overlap_active:      # a nested overlap on this device: runs on the outer one's current stream
                ov.on = False
        if not ov.on:
            return self
        # everything both passes read is formed now, on the main stream; what the caches do not know yet (the first step of a
        # model) is made lazily inside the region between two fences (_overlap_fence)
        if _wino_cache:
            _wino_refresh_all(dev)
        if _fold_cache:
            _fold_refresh_all(dev)
        from . import amp as _amp
        if _amp._cache:
            _amp._refresh_all(dev)
        ov.main = torch.cuda.current_stream(dev)
        key = (idx, ov.main.cuda_stream)
        side = _side_streams.get(key)
        if side is None:
            side = _side_streams[key] = torch.cuda.Stream(device=dev)
        ov.side = side
        side.wait_stream(ov.main)
        ov._key = idx
        _overlap_active[idx] = ov
        self.ctx = torch.cuda.stream(side)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


def begin_forward():
    """Top of a model forward: operands transformed before this point are not trusted any more."""
    _weights_epoch[0] += 1


class model_scope:
    """``with model_scope():`` around a composite forward (EncoderDecoder): the sub-modules' own ``begin_forward``
    calls are suppressed so the whole model costs ONE refresh launch."""

    def __enter__(self):
        if _scope_depth[0] == 0:
            begin_forward()
        _scope_depth[0] += 1

    def __exit__(self, *a):
        _scope_depth[0] -= 1
        if _scope_depth[0] == 0 and _overlap_active:    # (an overlap used without its ``with`` block and left through an exception)
            for ov in list(_overlap_active.values()):
                ov.join()
        if _scope_depth[0] == 0 and a and a[0] is not None:
            # a forward that raised half way: the BatchNorms it passed have updated their running statistics - apply their
            # ``num_batches_tracked`` increments too instead of leaving them queued for whatever forward comes next
            from . import functional as _SF
            _SF.flush_counters()
        return False


def begin_forward_unless_scoped():
    if _scope_depth[0] == 0:
        begin_forward()


def weights_changed():
    """Hint from the fused optimiser / EMA kernels; not needed for correctness (see above)."""
    _weights_epoch[0] += 1


class _WinoEntry:
    __slots__ = ("ref", "tag", "u", "shape")

    def __init__(self, w):
        Cout, Cin, _ = w.shape
        self.ref = weakref.ref(w)
        self.tag = None
        self.shape = (Cout, Cin, _wino_variant(Cout, Cin))
        planes = 3 if self.shape[2] == 4 else 4     # F(4,3): tap-major re-layout of the raw taps; F(2,3): four transformed planes
        # [forward operand, data-gradient operand]; both refreshed together by the multi-tensor launch
        self.u = [torch.empty((planes * Cout * Cin,), device=w.device, dtype=torch.float32) for _ in range(2)]


_wino_table = {}   # per variant: (key tuple, device table)  (table_for)


def _wino_refresh_all(device):
    """One launch per Winograd variant re-transforms every registered, still-alive weight (both orientations) and stamps it
    with the epoch."""
    _overlap_fence(device)
    try:
        _wino_refresh_all_unfenced(device)
    finally:
        _overlap_fence(device)


def _wino_refresh_all_unfenced(device):
    live = {2: [], 4: []}
    for key, ent in list(_wino_cache.items()):
        w = ent.ref()
        if w is None or w.data_ptr() != key or w.device != device or ent.shape[2] != _wino_variant(ent.shape[0], ent.shape[1]):
            if w is None:
                del _wino_cache[key]
            continue
        live[ent.shape[2]].append((w, ent))
    for var, items in live.items():
        if not items:
            continue
        rows = []
        for w, ent in items:
            rows += [w.data_ptr(), ent.u[0].data_ptr(), ent.u[1].data_ptr(), ent.shape[0], ent.shape[1]]
        key = tuple(rows)   # operand buffers included: a re-made entry has new ones
        tab = table_for(_wino_table, var, key, rows, device)
        for w, ent in items:
            keep_for_graph(w, ent.u[0], ent.u[1])
        fn = lib().ssecg_conv1d_wino4_weight_multi if var == 4 else lib().ssecg_conv1d_wino_weight_multi
        check(fn(_p(tab), len(items), max(e.shape[0] * e.shape[1] for _, e in items), _stream()), "ssecg_conv1d_wino_weight_multi")
        WINO_TRANSFORMS[0] += 1
        for w, ent in items:
            ent.tag = _weights_epoch[0]


def _wino_operand(w, transposed, cached=False):
    """Transformed weights.  ``cached=False``: transform now (single launch).  ``cached=True`` (model forward/backward):
    valid iff made from this very tensor object at this address in the current epoch; a stale entry refreshes EVERY
    registered weight in one multi-tensor launch."""
    key = w.data_ptr()
    ent = _wino_cache.get(key)
    if ent is None or ent.ref() is not w or ent.shape != (w.shape[0], w.shape[1], _wino_variant(w.shape[0], w.shape[1])):
        # unknown storage, or the entry was made from ANOTHER tensor object at this address (a freed tensor's successor,
        # or an alias such as the MeanTeacher teacher bound to the student's storage)
        if len(_wino_cache) > 512:
            _wino_cache.clear()
        ent = _wino_cache[key] = _WinoEntry(w)
    if not cached:
        if ent.shape[2] == 4:   # one-row table of the multi-tensor entry point (standalone calls are rare: tests, nn.Conv1d)
            row = [w.data_ptr(), 0 if transposed else ent.u[0].data_ptr(), ent.u[1].data_ptr() if transposed else 0, w.shape[0], w.shape[1]]
            tab = torch.tensor(row, dtype=torch.int64).to(w.device)
            check(lib().ssecg_conv1d_wino4_weight_multi(_p(tab), 1, w.shape[0] * w.shape[1], _stream()), "ssecg_conv1d_wino4_weight_multi")
        else:
            check(lib().ssecg_conv1d_wino_weight(_p(w), _p(ent.u[1 if transposed else 0]), w.shape[0], w.shape[1],
                                                 1 if transposed else 0, _stream()), "ssecg_conv1d_wino_weight")
        WINO_TRANSFORMS[0] += 1
        ent.tag = None          # the other orientation was not refreshed: cached users must not trust this entry
        return ent.u[1 if transposed else 0], ent.shape[2]
    if ent.tag != _weights_epoch[0]:
        _wino_refresh_all(w.device)
        if ent.tag != _weights_epoch[0]:
            raise SsecgError("internal: Winograd operand cache did not refresh")
    return ent.u[1 if transposed else 0], ent.shape[2]


def _wino_symbol(M, Q=1 << 30, var=2):
    """Kernel template instance the launcher picks (csrc/conv_wino.hip::pick_wino) - the name rocprofv3 reports."""
    if var == 4:
        return "conv_wino4_kernel<4, 2>" if M % 128 == 0 else "conv_wino4_kernel<2, 4>"
    bnp, bm = (128, 128) if M % 128 == 0 else (256, 64)   # small problems fall back to the 8-wave tiles
    wide = ((Q + bnp - 1) // bnp) * (M // bm) >= 256
    return f"conv_wino_kernel<{4 if M % 128 == 0 else 2}, {(4 if wide else 2) if M % 128 == 0 else (8 if wide else 4)}>"


def _wino_ok(N, C, L, M, K, stride, pad, dil):
    if not (WINOGRAD and K == 3 and stride == 1 and pad == 1 and dil == 1):
        return False
    return (lib().ssecg_conv1d_wino4_supported if _wino_variant(M, C) == 4 else lib().ssecg_conv1d_wino_supported)(N, C, L, M) == 1


def _conv1d_wino(src, w, transposed, scale, shift, residual, relu, want_stats, in_affine=None, w_cached=False):
    N, C, L = src.shape
    M = w.shape[1] if transposed else w.shape[0]
    Lb = lib()
    u, var = _wino_operand(w, transposed, w_cached)
    out = torch.empty((N, M, L), device=src.device, dtype=torch.float32)
    stats, parts = None, 0
    if want_stats:
        parts = (Lb.ssecg_conv1d_wino4_parts if var == 4 else Lb.ssecg_conv1d_wino_parts)(N, L, M)
        stats = torch.empty((parts, M, 2), device=src.device, dtype=torch.float32)
    trace("conv1d_wino", tuple(src.shape), M, "T" if transposed else "", "stats" if want_stats else "", "res" if residual is not None else "")
    with _Timed(_wino_symbol(M, N * ((L + 1) // 2), var), 2.0 * N * L * M * C * 3,
                4.0 * (N * C * L + N * M * L * (2 if residual is not None else 1) + 3 * M * C)):
        aff0, aff1 = (_p(in_affine[0]), _p(in_affine[1])) if in_affine else (None, None)
        if var == 4:
            # small launches (fewer tiles than CUs) contract their channels in up to 8 K splits side by side: workspace for the partials
            ws, nws = _split_ws("wino4", (N, C, L, M), src.device)
            check(Lb.ssecg_conv1d_wino4(_p(src), _p(u), _p(out), N, C, L, M, _p(scale), _p(shift), _p(residual), int(relu), _p(stats),
                                        parts, aff0, aff1, _p(ws), nws, _stream()), "ssecg_conv1d_wino4")
        else:
            check(Lb.ssecg_conv1d_wino(_p(src), _p(u), _p(out), N, C, L, M, _p(scale), _p(shift), _p(residual), int(relu), _p(stats),
                                       parts, aff0, aff1, _stream()), "ssecg_conv1d_wino")
    return out, stats


def _stem_ok(N, Cin, Lin, Cout, K, stride, pad, dil):
    return (STEM and Cout == 64 and K == 7 and stride == 2 and pad == 3 and dil == 1
            and lib().ssecg_stem_supported(N, Cin, Lin) == 1)


def stem_pair_ok(pair, w):
    a, b = pair
    N = a.shape[0] + b.shape[0]
    return (STEM and a.dtype == torch.float32 and a.is_contiguous() and b.is_contiguous()
            and _stem_ok(N, a.shape[1], a.shape[2], w.shape[0], w.shape[2], 2, 3, 1))


def stem_c16_ok(a, b=None) -> bool:
    """May the stem's conv output c and its gradient dc be STORED as bf16 (lp = 2) for input ``a`` [+ ``b``]?  The library's one answer
    for the forward AND the weight gradient (``ssecg_stem_c16_supported``) plus their 16-byte alignment of the inputs."""
    if not AMP_STEM_C16 or not isinstance(a, torch.Tensor) or a.dim() != 3:
        return False
    n = a.shape[0] + (0 if b is None else b.shape[0])
    if lib().ssecg_stem_c16_supported(n, a.shape[1], a.shape[2]) != 1:
        return False
    return a.data_ptr() % 16 == 0 and (b is None or b.data_ptr() % 16 == 0)


def stem_fwd_pair(pair, w, want_stats=True, lp=False):
    """The stem convolution over ``cat(pair)`` without the concatenated copy -> (c, stats_partial).  ``pair`` may also be ONE
    tensor (then ``lp`` is the reason to come here).  ``lp``: the use_amp form - x and w rounded to bf16 while staged, the output
    rounded to bf16 before the BatchNorm sums and the store (what autocast's 16-bit convolution computes; fp32 containers)."""
    single = isinstance(pair, torch.Tensor)
    a = _req(pair if single else pair[0], "x"); b = None if single else _req(pair[1], "x2"); w = _req(w, "w")
    Na, Cin, Lin = a.shape
    N = Na + (0 if b is None else b.shape[0])
    Cout, _, K = w.shape
    Lout = conv_out_len(Lin, K, 2, 3, 1)
    lp = int(lp)                        # 0 fp32; 1 bf16-rounded values in fp32 containers; 2 the same values stored as bf16 (planar)
    y = torch.empty((N, Cout, Lout), device=a.device, dtype=torch.bfloat16 if lp == 2 else torch.float32)
    L = lib()
    parts, stats = 0, None
    if want_stats:
        parts = L.ssecg_stem_parts(N, Lin)
        stats = torch.empty((parts, Cout, 2), device=a.device, dtype=torch.float32)
    trace("stem_fwd", (N, Cin, Lin), "pair" if b is not None else "", "stats" if want_stats else "", f"lp{lp}" if lp else "")
    with _Timed("stem_fwd_kernel<false>", 2.0 * N * Lout * Cout * Cin * K,
                4.0 * (a.numel() + (0 if b is None else b.numel())) + y.numel() * y.element_size()):
        check(L.ssecg_stem_fwd2(_p(a), _p(b), Na, _p(w), _p(y), N, Cin, Lin, _p(stats), parts, lp, _stream()), "ssecg_stem_fwd2")
    return y, stats


def stem_wgrad_pair(dy, pair, ksize=7, lp=False):
    """Weight gradient of the stem convolution over ``cat(pair)`` (or one tensor); ``lp``: x and dy rounded to bf16 while staged."""
    single = isinstance(pair, torch.Tensor)
    lp = int(lp)                        # 2: dy is stored as bf16 (bn_relu_maxpool_bwd_apply with lp = 2)
    dy = _req(dy, "dy", torch.bfloat16 if lp == 2 else torch.float32)
    a = _req(pair if single else pair[0], "x"); b = None if single else _req(pair[1], "x2")
    Na, Cin, Lin = a.shape
    N = Na + (0 if b is None else b.shape[0])
    Cout = dy.shape[1]
    L = lib()
    nbytes = L.ssecg_stem_wgrad_workspace(N, Cin, Lin)
    ws = _workspace(a.device, nbytes)
    dw = torch.empty((Cout, Cin, ksize), device=a.device, dtype=torch.float32)
    trace("stem_wgrad", tuple(dy.shape), (N, Cin, Lin), "pair" if b is not None else "", "ws", nbytes, f"lp{lp}" if lp else "")
    with _Timed("stem_wgrad_kernel + stem_wgrad_reduce_kernel", 2.0 * N * dy.shape[2] * Cout * Cin * ksize,
                dy.numel() * dy.element_size() + 4.0 * (a.numel() + (0 if b is None else b.numel()))):
        check(L.ssecg_stem_wgrad2(_p(dy), _p(a), _p(b), Na, _p(dw), N, Cin, Lin, _p(ws), ws.numel(), lp, _stream()),
              "ssecg_stem_wgrad2")
    return dw


def stem_fwd_eval_pool(x, w, scale, shift):
    """Eval-mode stem in one launch: maxpool_3,2,1(relu(conv_k7s2p3(x) * scale + shift)) -> (N, 64, Lp); None if the shape
    is not the stem's (the caller then chains conv1d_fwd + bn_relu_maxpool_fwd)."""
    x = _req(x, "x"); w = _req(w, "w")
    N, Cin, Lin = x.shape
    Cout, Cin2, K = w.shape
    if Cin2 != Cin or not _stem_ok(N, Cin, Lin, Cout, K, 2, 3, 1):
        return None
    Lout = (Lin - 1) // 2 + 1
    Lp = (Lout - 1) // 2 + 1
    y = torch.empty((N, 64, Lp), device=x.device, dtype=torch.float32)
    trace("stem_fwd_eval_pool", tuple(x.shape))
    with _Timed("stem_fwd_kernel<true>", 2.0 * N * Lout * 64 * Cin * 7, 4.0 * (x.numel() + y.numel())):
        check(lib().ssecg_stem_fwd_eval_pool(_p(x), _p(w), _p(_req(scale, "scale")), _p(_req(shift, "shift")), _p(y), N, Cin, Lin,
                                             _stream()), "ssecg_stem_fwd_eval_pool")
    return y


def conv1d_fwd(x, w, stride=1, pad=0, dil=1, scale=None, shift=None, residual=None, relu=False, want_stats=False,
               in_affine=None, w_cached=False):
    """-> (y, stats_partial or None).  See ssecg_conv1d_fwd in include/ssecg.h.
    ``in_affine`` = (scale, shift): the input is taken as relu(x*scale[c] + shift[c]) (fused producer BN + ReLU).
    ``w_cached``: the caller is inside a model forward/backward that called ``begin_forward`` (operand cache trusted)."""
    x = _req(x, "x"); w = _req(w, "w")
    N, Cin, Lin = x.shape
    Cout, Cin2, K = w.shape
    if Cin2 != Cin:
        raise SsecgError(f"conv1d: weight expects {Cin2} input channels, input has {Cin}")
    Lout = conv_out_len(Lin, K, stride, pad, dil)
    if scale is not None: scale = _req(scale, "scale")
    if shift is not None: shift = _req(shift, "shift")
    if residual is not None:
        residual = _req(residual, "residual")
        if tuple(residual.shape) != (N, Cout, Lout):
            raise SsecgError("conv1d: residual shape mismatch")
    if (in_affine is None or Cin <= 512) and _wino_ok(N, Cin, Lin, Cout, K, stride, pad, dil):
        return _conv1d_wino(x, w, False, scale, shift, residual, relu, want_stats, in_affine, w_cached)
    y = torch.empty((N, Cout, Lout), device=x.device, dtype=torch.float32)
    stats = None
    L = lib()
    if (scale is None and shift is None and residual is None and not relu and in_affine is None
            and _stem_ok(N, Cin, Lin, Cout, K, stride, pad, dil)):
        parts = 0
        if want_stats:
            parts = L.ssecg_stem_parts(N, Lin)
            stats = torch.empty((parts, Cout, 2), device=x.device, dtype=torch.float32)
        trace("stem_fwd", (N, Cin, Lin), "stats" if want_stats else "")
        with _Timed("stem_fwd_kernel<false>", 2.0 * N * Lout * Cout * Cin * K, 4.0 * (x.numel() + y.numel())):
            check(L.ssecg_stem_fwd(_p(x), _p(w), _p(y), N, Cin, Lin, _p(stats), parts, _stream()), "ssecg_stem_fwd")
        return y, stats
    parts = 0
    if want_stats:
        parts = L.ssecg_conv1d_stats_parts(N, Cin, Cout, Lout, K)
        stats = torch.empty((parts, Cout, 2), device=x.device, dtype=torch.float32)
    if scale is not None: scale = _req(scale, "scale")
    if shift is not None: shift = _req(shift, "shift")
    if residual is not None:
        residual = _req(residual, "residual")
        if residual.shape != y.shape:
            raise SsecgError("conv1d: residual shape mismatch")
    trace("conv1d_fwd", (N, Cin, Lin), (Cout, Cin, K), stride, pad, dil, "stats" if want_stats else "", "res" if residual is not None else "")
    with _Timed(_igemm_symbol(Cout, Cin, K, 0), 2.0 * N * Lout * Cout * Cin * K,
                4.0 * (N * Cin * Lin + N * Cout * Lout * (2 if residual is not None else 1) + Cout * Cin * K)):
        ws, nws = _split_ws("fwd", (N, Cin, Lin, Cout, Lout, K), x.device)
        check(L.ssecg_conv1d_fwd(_p(x), _p(w), _p(y), N, Cin, Lin, Cout, Lout, K, stride, pad, dil,
                                 _p(scale), _p(shift), _p(residual), int(relu), _p(stats), parts,
                                 _p(in_affine[0]) if in_affine else None, _p(in_affine[1]) if in_affine else None, _p(ws), nws,
                                 _stream()),
              "ssecg_conv1d_fwd")
    return y, stats


_split_query = {}
_ksplit_off = [0]


class ksplit_disabled:
    """``with ksplit_disabled():`` - no launch inside is K-split.  ``evaluate()`` / ``test.py`` / ``inference.py`` use it: the split
    decision depends on the number of position tiles, i.e. on the batch size, and changes the summation order - validation and test
    metrics must not flip near-tie arg-maxes with the dataloader's batch size.  Training passes keep the split."""

    def __enter__(self):
        _ksplit_off[0] += 1
        return self

    def __exit__(self, *exc):
        _ksplit_off[0] -= 1
        return False



def _split_ws(kind, shape, device):
    """Workspace of a K-split launch -> (tensor or None, bytes).  ``kind`` = "fwd" / "dgrad" / "wino4", ``shape`` = the arguments of
    the library's sizing query (cached per shape: the answer is a pure function of it).  The partial planes live in the per-(device,
    stream) scratch the weight gradients use too - each launch's finishing pass consumes them before the next launch on the
    stream can overwrite them."""
    if not KSPLIT or _ksplit_off[0]:
        return None, 0
    key = (kind,) + tuple(shape)
    nbytes = _split_query.get(key)
    if nbytes is None:
        L = lib()
        if kind == "wino4":
            N, C, Lx, M = shape
            S = L.ssecg_conv1d_wino4_split(N, C, Lx, M)
            nbytes = 4 * S * N * M * Lx if S > 1 else 0
        elif kind == "fwd":
            nbytes = int(L.ssecg_conv1d_fwd_split_workspace(*shape))
        else:
            nbytes = int(L.ssecg_conv1d_dgrad_split_workspace(*shape))
        _split_query[key] = nbytes
    if nbytes <= 0:
        return None, 0
    ws = _workspace(device, nbytes)
    return ws, ws.numel()


def conv1d_transpose_weight(w, stride=1):
    """Operand of the data-gradient GEMM (layout depends on the stride, see include/ssecg.h)."""
    trace("conv1d_transpose_weight", tuple(getattr(w, "shape", ())))
    w = _req(w, "w")
    Cout, Cin, K = w.shape
    wt = torch.empty((Cin, Cout, K), device=w.device, dtype=torch.float32)
    check(lib().ssecg_conv1d_transpose_weight(_p(w), _p(wt), Cout, Cin, K, stride, _stream()),
          "ssecg_conv1d_transpose_weight")
    return wt


def conv1d_dgrad(dy, w, in_len, stride=1, pad=0, dil=1, accumulate=None, w_cached=False, inplace=False):
    """dx of conv1d(x, w); ``w`` in the forward layout (Cout, Cin, K).  ``accumulate``: a gradient already computed for the same
    input, added in the epilogue; ``inplace``: the sum is written back into ``accumulate`` itself (direct kernels only)."""
    dy = _req(dy, "dy"); w = _req(w, "w")
    if accumulate is not None:
        accumulate = _req(accumulate, "accumulate")
        if tuple(accumulate.shape) != (dy.shape[0], w.shape[1], in_len):
            raise SsecgError("conv1d_dgrad: accumulate shape mismatch")
    if in_len == dy.shape[2] and _wino_ok(dy.shape[0], w.shape[0], in_len, w.shape[1], w.shape[2], stride, pad, dil):
        return _conv1d_wino(dy, w, True, None, None, accumulate, False, False, None, w_cached)[0]
    wt = conv1d_transpose_weight(w, stride)
    N, Cout, Lout = dy.shape
    Cin, _, K = wt.shape
    dx = accumulate if (inplace and accumulate is not None) else torch.empty((N, Cin, in_len), device=dy.device, dtype=torch.float32)
    trace("conv1d_dgrad", tuple(dy.shape), (Cout, Cin, K), in_len, stride, pad, dil, "acc" if accumulate is not None else "")
    with _Timed(_igemm_symbol(Cin, Cout, K, 1, stride), 2.0 * N * Lout * Cout * Cin * K,
                4.0 * (N * Cout * Lout + N * Cin * in_len * (2 if accumulate is not None else 1) + Cout * Cin * K)):
        ws, nws = _split_ws("dgrad", (N, Cin, in_len, Cout, Lout, K, stride), dy.device)
        check(lib().ssecg_conv1d_dgrad(_p(dy), _p(wt), _p(dx), N, Cin, in_len, Cout, Lout, K, stride, pad, dil,
                                       _p(accumulate), _p(ws), nws, _stream()), "ssecg_conv1d_dgrad")
    return dx


_ws_cache = {}


def _workspace(device, nbytes: int) -> torch.Tensor:
    """Per-device scratch reused by every wgrad launch on the stream (stream-ordered reuse is safe)."""
    key = (device, _stream())
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty((max(nbytes, 1 << 20),), device=device, dtype=torch.uint8)
        _ws_cache[key] = buf
    return buf


def conv1d_wgrad(dy, x, ksize, stride=1, pad=0, dil=1, x_affine=None):
    dy = _req(dy, "dy"); x = _req(x, "x")
    N, Cout, Lout = dy.shape
    _, Cin, Lin = x.shape
    L = lib()
    f4 = WINO_WGRAD_F == 4
    if (WINOGRAD and ksize == 3 and stride == 1 and pad == 1 and dil == 1 and Lin == Lout
            and (L.ssecg_conv1d_wino_wgrad4_supported if f4 else L.ssecg_conv1d_wino_wgrad_supported)(N, Cin, Lin, Cout) == 1):
        nbytes = (L.ssecg_conv1d_wino_wgrad4_workspace if f4 else L.ssecg_conv1d_wino_wgrad_workspace)(N, Cin, Lin, Cout)
        ws = _workspace(x.device, nbytes)
        dw = torch.empty((Cout, Cin, 3), device=x.device, dtype=torch.float32)
        trace("conv1d_wino_wgrad4" if f4 else "conv1d_wino_wgrad", tuple(dy.shape), tuple(x.shape), "ws", nbytes)
        with _Timed("conv_wino_wgrad4_kernel + wino_wgrad4_reduce_kernel" if f4 else "conv_wino_wgrad_kernel + wino_wgrad_reduce_kernel",
                    2.0 * N * Lout * Cout * Cin * 3, 4.0 * (N * Cout * Lout + N * Cin * Lin + Cout * Cin * 3)):
            check((L.ssecg_conv1d_wino_wgrad4 if f4 else L.ssecg_conv1d_wino_wgrad)(
                _p(dy), _p(x), _p(dw), N, Cin, Lin, Cout, _p(ws), ws.numel(), _p(x_affine[0]) if x_affine else None,
                _p(x_affine[1]) if x_affine else None, _stream()), "ssecg_conv1d_wino_wgrad4" if f4 else "ssecg_conv1d_wino_wgrad")
        return dw
    if x_affine is None and _stem_ok(N, Cin, Lin, Cout, ksize, stride, pad, dil) and Lout == (Lin - 1) // 2 + 1:
        nbytes = L.ssecg_stem_wgrad_workspace(N, Cin, Lin)
        ws = _workspace(x.device, nbytes)
        dw = torch.empty((Cout, Cin, ksize), device=x.device, dtype=torch.float32)
        trace("stem_wgrad", tuple(dy.shape), tuple(x.shape), "ws", nbytes)
        with _Timed("stem_wgrad_kernel + stem_wgrad_reduce_kernel", 2.0 * N * Lout * Cout * Cin * ksize,
                    4.0 * (dy.numel() + x.numel())):
            check(L.ssecg_stem_wgrad(_p(dy), _p(x), _p(dw), N, Cin, Lin, _p(ws), ws.numel(), _stream()), "ssecg_stem_wgrad")
        return dw
    nbytes = L.ssecg_conv1d_wgrad_workspace(N, Cin, Lin, Cout, Lout, ksize)
    ws = _workspace(x.device, nbytes)
    dw = torch.empty((Cout, Cin, ksize), device=x.device, dtype=torch.float32)
    trace("conv1d_wgrad", tuple(dy.shape), tuple(x.shape), ksize, stride, pad, dil, "ws", nbytes)
    with _Timed("conv_wgrad_kernel + wgrad_reduce_kernel", 2.0 * N * Lout * Cout * Cin * ksize,
                4.0 * (N * Cout * Lout + N * Cin * Lin + Cout * Cin * ksize)):
        check(L.ssecg_conv1d_wgrad(_p(dy), _p(x), _p(dw), N, Cin, Lin, Cout, Lout, ksize, stride, pad, dil,
                                   _p(ws), ws.numel(), _p(x_affine[0]) if x_affine else None,
                                   _p(x_affine[1]) if x_affine else None, _stream()), "ssecg_conv1d_wgrad")
    return dw


# ----------------------------------------------------------------------------- batch norm
def bn_reduce_partials(partial, want_param_grads=False, out=None):
    """-> sums (C,2) f64 [, dgamma, dbeta].  ``out``: C rows of a larger (rows, 2) f64 buffer - two BatchNorms whose sums
    travel in ONE all-reduce write into neighbouring row ranges of it."""
    trace("bn_reduce_partials", tuple(getattr(partial, "shape", ())))
    partial = _req(partial, "partial")
    parts, C, _ = partial.shape
    if out is not None:
        sums = _req(out, "out", torch.float64)
        if tuple(sums.shape) != (C, 2):
            raise SsecgError("bn_reduce_partials: out must be (C, 2) float64")
    else:
        sums = torch.empty((C, 2), device=partial.device, dtype=torch.float64)
    dg = db = None
    if want_param_grads:
        dg = torch.empty((C,), device=partial.device, dtype=torch.float32)
        db = torch.empty((C,), device=partial.device, dtype=torch.float32)
    check(lib().ssecg_bn_reduce_partials(_p(partial), parts, C, _p(sums), _p(dg), _p(db), _stream()),
          "ssecg_bn_reduce_partials")
    return (sums, dg, db) if want_param_grads else sums


def bn_stats_finalize(partial, count, eps, momentum, running_mean=None, running_var=None, affine_of=None):
    """Single-GPU train-mode statistics in one launch -> (mean, invstd[, (scale, shift)]); running stats updated in
    place.  ``affine_of`` = (gamma, beta): also emit the per-channel affine a consumer conv applies in its gather."""
    trace("bn_stats_finalize", tuple(getattr(partial, "shape", ())))
    partial = _req(partial, "partial")
    parts, C, _ = partial.shape
    mean = torch.empty((C,), device=partial.device, dtype=torch.float32)
    invstd = torch.empty((C,), device=partial.device, dtype=torch.float32)
    aff = torch.empty((2, C), device=partial.device, dtype=torch.float32) if affine_of is not None else None
    check(lib().ssecg_bn_stats_finalize(_p(partial), parts, C, float(count), float(eps), float(momentum), _p(mean),
                                        _p(invstd), _p(running_mean), _p(running_var),
                                        _p(affine_of[0]) if affine_of else None, _p(affine_of[1]) if affine_of else None,
                                        aff[0].data_ptr() if aff is not None else None,
                                        aff[1].data_ptr() if aff is not None else None, _stream()), "ssecg_bn_stats_finalize")
    return (mean, invstd, (aff[0], aff[1])) if aff is not None else (mean, invstd)


def bn_finalize(sums, count, eps, momentum, running_mean=None, running_var=None, affine_of=None):
    trace("bn_finalize", tuple(getattr(sums, "shape", ())))
    C = sums.shape[0]
    mean = torch.empty((C,), device=sums.device, dtype=torch.float32)
    invstd = torch.empty((C,), device=sums.device, dtype=torch.float32)
    aff = torch.empty((2, C), device=sums.device, dtype=torch.float32) if affine_of is not None else None
    check(lib().ssecg_bn_finalize(_p(sums), C, float(count), float(eps), float(momentum), _p(mean), _p(invstd),
                                  _p(running_mean), _p(running_var),
                                  _p(affine_of[0]) if affine_of else None, _p(affine_of[1]) if affine_of else None,
                                  aff[0].data_ptr() if aff is not None else None,
                                  aff[1].data_ptr() if aff is not None else None, _stream()), "ssecg_bn_finalize")
    return (mean, invstd, (aff[0], aff[1])) if aff is not None else (mean, invstd)


def bn_fold(gamma, beta, running_mean, running_var, eps):
    trace("bn_fold", tuple(getattr(gamma, "shape", ())))
    gamma = _req(gamma, "gamma"); beta = _req(beta, "beta")
    running_mean = _req(running_mean, "running_mean"); running_var = _req(running_var, "running_var")
    C = gamma.shape[0]
    out = torch.empty((2, C), device=gamma.device, dtype=torch.float32)
    check(lib().ssecg_bn_fold(_p(gamma), _p(beta), _p(running_mean), _p(running_var), C, float(eps),
                              out[0].data_ptr(), out[1].data_ptr(), _stream()), "ssecg_bn_fold")
    return out[0], out[1]


class _FoldEntry:
    __slots__ = ("refs", "tag", "out", "eps")

    def __init__(self, gamma, beta, rm, rv, eps):
        self.refs = tuple(weakref.ref(t) for t in (gamma, beta, rm, rv))
        self.tag = None
        self.eps = float(eps)
        self.out = torch.empty((2, gamma.shape[0]), device=gamma.device, dtype=torch.float32)


_fold_cache = {}
_fold_table = {}
FOLD_LAUNCHES = [0]


def _fold_refresh_all(device):
    _overlap_fence(device)
    try:
        _fold_refresh_all_unfenced(device)
    finally:
        _overlap_fence(device)


def _fold_refresh_all_unfenced(device):
    import struct
    rows, live, mx = [], [], 1
    for key, ent in list(_fold_cache.items()):
        ts = [r() for r in ent.refs]
        if any(t is None for t in ts) or ts[2].data_ptr() != key or ts[0].device != device:
            if any(t is None for t in ts):
                del _fold_cache[key]
            continue
        C = ts[0].shape[0]
        rows += [ts[0].data_ptr(), ts[1].data_ptr(), ts[2].data_ptr(), ts[3].data_ptr(), ent.out[0].data_ptr(), ent.out[1].data_ptr(),
                 C, struct.unpack("<i", struct.pack("<f", ent.eps))[0]]
        mx = max(mx, C)
        live.append(ent)
        keep_for_graph(*ts, ent.out[0], ent.out[1])
    if not live:       # every registered BatchNorm has been freed (the entries above are gone now)
        return
    key = tuple(rows)
    tab = table_for(_fold_table, 0, key, rows, device)
    check(lib().ssecg_bn_fold_multi(_p(tab), len(live), mx, _stream()), "ssecg_bn_fold_multi")
    FOLD_LAUNCHES[0] += 1
    for ent in live:
        ent.tag = _weights_epoch[0]


def bn_fold_cached(gamma, beta, running_mean, running_var, eps):
    """``bn_fold`` for the eval-mode passes of a model (teacher / pseudo-label pass, evaluate()): the first request after
    ``begin_forward()`` folds EVERY BatchNorm registered so far in ONE launch (21 launches -> 1 per pass); an entry is
    trusted only if it was made from these very tensor objects in the current epoch - the rule of the Winograd operands."""
    gamma = _req(gamma, "gamma"); beta = _req(beta, "beta")
    running_mean = _req(running_mean, "running_mean"); running_var = _req(running_var, "running_var")
    key = running_mean.data_ptr()
    ent = _fold_cache.get(key)
    if (ent is None or any(r() is not t for r, t in zip(ent.refs, (gamma, beta, running_mean, running_var)))
            or ent.eps != float(eps) or ent.out.shape[1] != gamma.shape[0]):
        if len(_fold_cache) > 512:
            _fold_cache.clear()
        ent = _fold_cache[key] = _FoldEntry(gamma, beta, running_mean, running_var, eps)
    if ent.tag != _weights_epoch[0]:
        _fold_refresh_all(gamma.device)
        if ent.tag != _weights_epoch[0]:
            raise SsecgError("internal: BatchNorm fold cache did not refresh")
    return ent.out[0], ent.out[1]


#: the ReLU mask of a unit whose ReLU follows a residual add travels to the backward as PACKED BITS written by the forward apply
#: pass (1/32 of the saved activation's bytes in both backward passes); SSECG_BN_MASK_BITS=0 reads the saved activation instead.
config.passthrough("SSECG_BN_ROWS", "set (any value): BatchNorm backward reduction of rows with L % 4 != 0 on the 4-byte path of rounds 1-5 instead of "
                                    "the 16-byte raw-buffer-load rows kernel (read by csrc/elementwise.hip per call)")
BN_MASK_BITS = config.switch("SSECG_BN_MASK_BITS", True, "ReLU masks of residual units travel as packed bits", __name__, "BN_MASK_BITS")


def bn_mask_supported(N, C, L) -> bool:
    return BN_MASK_BITS and lib().ssecg_bn_mask_supported(int(N), int(C), int(L)) == 1


def bn_apply_fwd(x, mean, invstd, gamma, beta, residual=None, relu=False, want_mask=False, res_bn=None):
    """-> y, or (y, mask) with ``want_mask``: ``mask`` = uint8 (ceil(numel / 8),), bit (e & 7) of byte (e >> 3) = (y[e] > 0).
    ``res_bn`` = (mean, invstd, gamma, beta): ``residual`` is the RAW output of the block's 1x1 downsample convolution and that
    branch's BatchNorm is applied while it is read (the normalised identity tensor is never written; ``ssecg_bn_apply_fwd_resbn``)."""
    trace("bn_apply_fwd", tuple(getattr(x, "shape", ())))
    x = _req(x, "x")
    N, C, L = x.shape
    y = torch.empty_like(x)
    if residual is not None:
        residual = _req(residual, "residual")
        if tuple(residual.shape) != (N, C, L):
            raise SsecgError("bn_apply_fwd: residual shape mismatch")
    rb = [None] * 4 if res_bn is None else [_req(t, "res_bn") for t in res_bn]
    mask = torch.empty(((x.numel() + 7) // 8,), device=x.device, dtype=torch.uint8) if want_mask else None
    with _Timed("bn_apply_fwd_kernel", 0.0, 4.0 * x.numel() * (3 if residual is not None else 2)):
        check(lib().ssecg_bn_apply_fwd_resbn(_p(x), _p(y), N, C, L, _p(mean), _p(invstd), _p(_req(gamma, "gamma")),
                                             _p(_req(beta, "beta")), _p(residual), _p(rb[0]), _p(rb[1]), _p(rb[2]), _p(rb[3]), int(relu),
                                             _p(mask), _stream()), "ssecg_bn_apply_fwd_resbn")
    return (y, mask) if want_mask else y


def _mask_args(y):
    """``y`` of the backward passes: the saved activation (fp32, same shape) or the packed mask (uint8) -> (y ptr, bits ptr, bytes/el)."""
    if y is None:
        return None, None, 0.0
    if y.dtype == torch.uint8:
        if not y.is_cuda or not y.is_contiguous():
            raise SsecgError("mask: expected a contiguous HIP device tensor")
        return None, y.data_ptr(), 0.125
    return _req(y, "y").data_ptr(), None, 4.0


def bn_bwd_reduce(dy, y, x, mean, invstd, gamma=None, beta=None, relu_recompute=False):
    """ReLU mask from ``y`` (saved activation, or the packed uint8 mask of bn_apply_fwd) or, with ``relu_recompute``, recomputed
    from the BN input ``x``."""
    trace("bn_bwd_reduce", tuple(getattr(dy, "shape", ())))
    dy = _req(dy, "dy"); x = _req(x, "x")
    N, C, L = x.shape
    Lb = lib()
    parts = Lb.ssecg_bn_bwd_parts(N, C, L)
    partial = torch.empty((parts, C, 2), device=x.device, dtype=torch.float32)
    yp, bp, yb = _mask_args(y)
    with _Timed("bn_bwd_reduce_kernel", 0.0, x.numel() * (8.0 + yb)):
        check(Lb.ssecg_bn_bwd_reduce(_p(dy), yp, _p(x), _p(mean), _p(invstd), _p(gamma), _p(beta), int(relu_recompute),
                                     N, C, L, _p(partial), bp, _stream()), "ssecg_bn_bwd_reduce")
    return partial


def bn_bwd_apply(dy, y, x, mean, invstd, gamma, sums, count, want_dz=False, beta=None, relu_recompute=False):
    trace("bn_bwd_apply", tuple(getattr(dy, "shape", ())))
    dy = _req(dy, "dy"); x = _req(x, "x")
    N, C, L = x.shape
    dx = torch.empty_like(x)
    dz = torch.empty_like(x) if want_dz else None
    yp, bp, yb = _mask_args(y)
    with _Timed("bn_bwd_apply_kernel", 0.0, x.numel() * (8.0 + yb + (8.0 if want_dz else 4.0))):
        check(lib().ssecg_bn_bwd_apply(_p(dy), yp, _p(x), _p(mean), _p(invstd), _p(_req(gamma, "gamma")), _p(beta),
                                       int(relu_recompute), _p(sums), float(count), N, C, L, _p(dx), _p(dz), bp, _stream()),
              "ssecg_bn_bwd_apply")
    return dx, dz


def bn_bwd_pair_supported(N, C, L) -> bool:
    return lib().ssecg_bn_bwd_pair_supported(int(N), int(C), int(L)) == 1


def bn_bwd_reduce_pair(dy, y, x, mean, invstd, x2, mean2, invstd2):
    """``bn_bwd_reduce`` for TWO BatchNorms behind one masked gradient (a downsample block's bn2 and the BatchNorm of its 1x1 branch):
    ``dy`` and the block's ReLU mask ``y`` (saved activation or packed uint8 bits) are read once -> (partial, partial2), each what
    the single launch writes, bit for bit."""
    trace("bn_bwd_reduce_pair", tuple(getattr(dy, "shape", ())))
    dy = _req(dy, "dy"); x = _req(x, "x"); x2 = _req(x2, "x2")
    N, C, L = x.shape
    if tuple(x2.shape) != (N, C, L) or tuple(dy.shape) != (N, C, L):
        raise SsecgError("bn_bwd_reduce_pair: shape mismatch")
    Lb = lib()
    parts = Lb.ssecg_bn_bwd_parts(N, C, L)
    partial = torch.empty((parts, C, 2), device=x.device, dtype=torch.float32)
    partial2 = torch.empty((parts, C, 2), device=x.device, dtype=torch.float32)
    yp, bp, yb = _mask_args(y)
    with _Timed("bn_bwd_reduce_kernel (pair)", 0.0, x.numel() * (12.0 + yb)):
        check(Lb.ssecg_bn_bwd_reduce_pair(_p(dy), yp, bp, _p(x), _p(mean), _p(invstd), _p(x2), _p(mean2), _p(invstd2), N, C, L,
                                          _p(partial), _p(partial2), _stream()), "ssecg_bn_bwd_reduce_pair")
    return partial, partial2


def bn_bwd_apply_pair(dy, y, x, mean, invstd, gamma, sums, x2, mean2, invstd2, gamma2, sums2, count):
    """``bn_bwd_apply`` for the same pair -> (dx, dx2); no dz is written (both consumers of it are in this pass)."""
    trace("bn_bwd_apply_pair", tuple(getattr(dy, "shape", ())))
    dy = _req(dy, "dy"); x = _req(x, "x"); x2 = _req(x2, "x2")
    N, C, L = x.shape
    dx, dx2 = torch.empty_like(x), torch.empty_like(x2)
    yp, bp, yb = _mask_args(y)
    with _Timed("bn_bwd_apply_kernel (pair)", 0.0, x.numel() * (20.0 + yb)):
        check(lib().ssecg_bn_bwd_apply_pair(_p(dy), yp, bp, _p(x), _p(mean), _p(invstd), _p(_req(gamma, "gamma")), _p(sums), _p(x2),
                                            _p(mean2), _p(invstd2), _p(_req(gamma2, "gamma2")), _p(sums2), float(count), N, C, L,
                                            _p(dx), _p(dx2), _stream()), "ssecg_bn_bwd_apply_pair")
    return dx, dx2


def bn_param_grads(sums):
    C = sums.shape[0]
    dg = torch.empty((C,), device=sums.device, dtype=torch.float32)
    db = torch.empty((C,), device=sums.device, dtype=torch.float32)
    check(lib().ssecg_bn_param_grads(_p(sums), C, _p(dg), _p(db), _stream()), "ssecg_bn_param_grads")
    return dg, db


def channel_sum(x):
    trace("channel_sum", tuple(getattr(x, "shape", ())))
    x = _req(x, "x")
    N, C, L = x.shape
    out = torch.empty((C,), device=x.device, dtype=torch.float32)
    ws = _workspace(x.device, 4 * C * 32)      # per (device, stream): two streams never share slab sums
    check(lib().ssecg_channel_sum(_p(x), N, C, L, _p(out), _p(ws), ws.numel(), _stream()), "ssecg_channel_sum")
    return out


# ----------------------------------------------------------------------------- pool / interp / dropout
def maxpool1d_fwd(x, k=3, stride=2, pad=1):
    trace("maxpool1d_fwd", tuple(getattr(x, "shape", ())))
    x = _req(x, "x")
    N, C, Lin = x.shape
    Lout = (Lin + 2 * pad - k) // stride + 1
    y = torch.empty((N, C, Lout), device=x.device, dtype=torch.float32)
    check(lib().ssecg_maxpool1d_fwd(_p(x), _p(y), N * C, Lin, Lout, k, stride, pad, _stream()), "ssecg_maxpool1d_fwd")
    return y


def maxpool1d_bwd(x, dy, k=3, stride=2, pad=1):
    trace("maxpool1d_bwd", tuple(getattr(x, "shape", ())))
    x = _req(x, "x"); dy = _req(dy, "dy")
    N, C, Lin = x.shape
    dx = torch.empty_like(x)
    check(lib().ssecg_maxpool1d_bwd(_p(x), _p(dy), _p(dx), N * C, Lin, dy.shape[2], k, stride, pad, _stream()),
          "ssecg_maxpool1d_bwd")
    return dx


def bn_relu_maxpool_fwd(x, mean, invstd, gamma, beta, k=3, stride=2, pad=1):
    """maxpool(relu(bn(x))) in one pass; eval mode: mean = invstd = None, gamma/beta = folded scale/shift."""
    x = _req(x, "x")
    N, C, Lin = x.shape
    Lout = (Lin + 2 * pad - k) // stride + 1
    y = torch.empty((N, C, Lout), device=x.device, dtype=torch.float32)
    trace("bn_relu_maxpool_fwd", tuple(x.shape))
    with _Timed("bn_relu_maxpool_fwd_kernel", 0.0, 4.0 * (x.numel() + y.numel())):
        check(lib().ssecg_bn_relu_maxpool_fwd(_p(x), _p(y), N, C, Lin, Lout, k, stride, pad, _p(mean), _p(invstd),
                                              _p(_req(gamma, "gamma")), _p(_req(beta, "beta")), _stream()),
              "ssecg_bn_relu_maxpool_fwd")
    return y


def bn_relu_maxpool_bwd_reduce(dy, x, mean, invstd, gamma, beta, k=3, stride=2, pad=1, lp=False):
    lp = int(lp)                        # 1: route on the bf16-rounded activation; 2: ... and x is stored as bf16
    dy = _req(dy, "dy"); x = _req(x, "x", torch.bfloat16 if lp == 2 else torch.float32)
    N, C, Lin = x.shape
    Lb = lib()
    parts = Lb.ssecg_bn_bwd_parts(N, C, Lin)
    partial = torch.empty((parts, C, 2), device=x.device, dtype=torch.float32)
    trace("bn_relu_maxpool_bwd_reduce", tuple(x.shape))
    with _Timed("bn_relu_maxpool_bwd_reduce_kernel", 0.0, x.numel() * x.element_size() + 4.0 * dy.numel()):
        check(Lb.ssecg_bn_relu_maxpool_bwd_reduce(_p(dy), _p(x), _p(mean), _p(invstd), _p(gamma), _p(beta), N, C, Lin,
                                                  dy.shape[2], k, stride, pad, _p(partial), lp, _stream()),
              "ssecg_bn_relu_maxpool_bwd_reduce")
    return partial


def bn_relu_maxpool_bwd_apply(dy, x, mean, invstd, gamma, beta, sums, count, k=3, stride=2, pad=1, lp=False):
    lp = int(lp)                        # 2: x is stored as bf16 and dx is written as bf16
    dy = _req(dy, "dy"); x = _req(x, "x", torch.bfloat16 if lp == 2 else torch.float32)
    N, C, Lin = x.shape
    dx = torch.empty_like(x)
    trace("bn_relu_maxpool_bwd_apply", tuple(x.shape))
    with _Timed("bn_relu_maxpool_bwd_apply_kernel", 0.0, 2 * x.numel() * x.element_size() + 4.0 * dy.numel()):
        check(lib().ssecg_bn_relu_maxpool_bwd_apply(_p(dy), _p(x), _p(mean), _p(invstd), _p(gamma), _p(beta), _p(sums),
                                                    float(count), N, C, Lin, dy.shape[2], k, stride, pad, _p(dx), lp,
                                                    _stream()),
              "ssecg_bn_relu_maxpool_bwd_apply")
    return dx


# ---- the stem's BN + ReLU + MaxPool writing the pooled activation in blocked bf16 (use_amp: no fp32 pooled tensor)
def stem_pool_b16_supported(N, C, Lin):
    return lib().ssecg_amp_stem_pool_supported(N, C, Lin) == 1


def stem_pool_fwd_b16(x, mean, invstd, gamma, beta):
    """maxpool(relu(bn(x)), 3, 2, 1) -> (N, C/8, Lin/2, 8) bf16; bit-identical to bn_relu_maxpool_fwd + amp.to_blocked."""
    x16 = isinstance(x, torch.Tensor) and x.dtype == torch.bfloat16      # the conv output stored as bf16 (stem_fwd_pair with lp = 2)
    x = _req(x, "x", torch.bfloat16 if x16 else torch.float32)
    N, C, Lin = x.shape
    y = torch.empty((N, C // 8, Lin // 2, 8), device=x.device, dtype=torch.bfloat16)
    trace("stem_pool_fwd_b16", tuple(x.shape))
    with _Timed("stem_pool_fwd_b16_kernel", 0.0, x.numel() * x.element_size() + 2.0 * y.numel()):
        check(lib().ssecg_amp_stem_pool_fwd(_p(x), _p(y), N, C, Lin, _p(mean), _p(invstd), _p(_req(gamma, "gamma")), _p(_req(beta, "beta")),
                                            int(x16), _stream()), "ssecg_amp_stem_pool_fwd")
    return y


def interp_linear_fwd(x, size, align_corners=False):
    trace("interp_linear_fwd", tuple(getattr(x, "shape", ())))
    x = _req(x, "x")
    N, C, Lin = x.shape
    y = torch.empty((N, C, size), device=x.device, dtype=torch.float32)
    with _Timed("interp_linear_fwd_kernel", 0.0, 4.0 * (x.numel() + y.numel())):
        check(lib().ssecg_interp_linear_fwd(_p(x), _p(y), N * C, Lin, size, int(bool(align_corners)), _stream()),
              "ssecg_interp_linear_fwd")
    return y


def interp_linear_bwd(dy, in_len, align_corners=False):
    trace("interp_linear_bwd", tuple(getattr(dy, "shape", ())))
    dy = _req(dy, "dy")
    N, C, Lout = dy.shape
    dx = torch.empty((N, C, in_len), device=dy.device, dtype=torch.float32)
    with _Timed("interp_linear_bwd_kernel", 0.0, 4.0 * (dy.numel() + dx.numel())):
        check(lib().ssecg_interp_linear_bwd(_p(dy), _p(dx), N * C, in_len, Lout, int(bool(align_corners)), _stream()),
              "ssecg_interp_linear_bwd")
    return dx


def draw_seed() -> int:
    """One dropout seed from torch's CPU generator (what the heads draw per train-mode forward; also the refresh hook of a seed
    slot in a captured step - the same draw in the same order as the eager step)."""
    return int(torch.randint(0, 2 ** 62, (1,)).item())


def dropout_fwd(x, p, seed):
    trace("dropout_fwd", tuple(getattr(x, "shape", ())))
    x = _req(x, "x")
    y = torch.empty_like(x)
    mask = torch.empty(x.shape, device=x.device, dtype=torch.uint8)
    seed_dev = STEP_SCALARS.seed_slot(seed) if STEP_SCALARS is not None else None   # HIP-graph capture: the seed lives on the device
    check(lib().ssecg_dropout_fwd(_p(x), _p(y), _p(mask), x.numel(), float(p), int(seed) & (2**64 - 1), seed_dev, _stream()),
          "ssecg_dropout_fwd")
    return y, mask


def mask_scale(x, mask, scale):
    trace("mask_scale", tuple(getattr(x, "shape", ())))
    x = _req(x, "x"); mask = _req(mask, "mask", torch.uint8)
    if mask.numel() != x.numel():
        raise SsecgError("mask_scale: mask shape mismatch")
    y = torch.empty_like(x)
    check(lib().ssecg_mask_scale(_p(x), _p(mask), _p(y), x.numel(), float(scale), _stream()), "ssecg_mask_scale")
    return y


# ----------------------------------------------------------------------------- pseudo labels / losses
def softmax_conf_argmax(logits, want_prob=False):
    """-> (conf (N,L) f32, mask (N,L) i64, prob (N,K,L) f32 or None)."""
    trace("softmax_conf_argmax", tuple(getattr(logits, "shape", ())))
    logits = _req(logits, "logits")
    N, K, L = logits.shape
    conf = torch.empty((N, L), device=logits.device, dtype=torch.float32)
    mask = torch.empty((N, L), device=logits.device, dtype=torch.int64)
    prob = torch.empty_like(logits) if want_prob else None
    with _Timed("softmax_conf_argmax_kernel", 0.0, 4.0 * logits.numel() * (2 if want_prob else 1) + 12.0 * N * L):
        check(lib().ssecg_softmax_conf_argmax(_p(logits), N, K, L, _p(conf), _p(mask), _p(prob), _stream()),
              "ssecg_softmax_conf_argmax")
    return conf, mask, prob


def seg_confusion(pred, target, num_classes):
    """-> counts (N, K, K) int32, counts[n, t, p] = #{l: target[n,l]==t, pred[n,l]==p}."""
    trace("seg_confusion", tuple(getattr(pred, "shape", ())))
    pred = _req(pred, "pred", torch.int64); target = _req(target, "target", torch.int64)
    if pred.dim() != 2 or tuple(pred.shape) != tuple(target.shape):
        raise SsecgError("seg_confusion: pred and target must both be (N, L)")
    N, L = pred.shape
    counts = torch.empty((N, num_classes, num_classes), device=pred.device, dtype=torch.int32)
    check(lib().ssecg_seg_confusion(_p(pred), _p(target), N, int(num_classes), L, _p(counts), _stream()), "ssecg_seg_confusion")
    return counts


def ce_hard_fwd_bwd(logits, target, conf=None, thresh=0.0, grad_scale=1.0, dlogits=None):
    """-> (dlogits, partial[parts,2] = {sum loss, sum weight})."""
    trace("ce_hard_fwd_bwd", tuple(getattr(logits, "shape", ())))
    logits = _req(logits, "logits"); target = _req(target, "target", torch.int64)
    N, K, L = logits.shape
    if tuple(target.shape) != (N, L):
        raise SsecgError("ce_hard: target must be (N, L)")
    if conf is not None:
        conf = _req(conf, "conf")
    Lb = lib()
    parts = Lb.ssecg_ce_parts(N, L)
    partial = torch.empty((parts, 2), device=logits.device, dtype=torch.float32)
    if dlogits is None:
        dlogits = torch.empty_like(logits)
    with _Timed("ce_hard_fwd_bwd_kernel", 0.0, 8.0 * logits.numel() + (12.0 if conf is not None else 8.0) * N * L):
        check(Lb.ssecg_ce_hard_fwd_bwd(_p(logits), _p(target), _p(conf), float(thresh), N, K, L, float(grad_scale),
                                       _p(dlogits), _p(partial), _stream()), "ssecg_ce_hard_fwd_bwd")
    return dlogits, partial


def ce_soft_fwd_bwd(logits, prob, grad_scale=1.0, dlogits=None):
    trace("ce_soft_fwd_bwd", tuple(getattr(logits, "shape", ())))
    logits = _req(logits, "logits"); prob = _req(prob, "prob")
    N, K, L = logits.shape
    Lb = lib()
    parts = Lb.ssecg_ce_parts(N, L)
    partial = torch.empty((parts, 2), device=logits.device, dtype=torch.float32)
    if dlogits is None:
        dlogits = torch.empty_like(logits)
    with _Timed("ce_soft_fwd_bwd_kernel", 0.0, 12.0 * logits.numel()):
        check(Lb.ssecg_ce_soft_fwd_bwd(_p(logits), _p(prob), N, K, L, float(grad_scale), _p(dlogits), _p(partial), _stream()),
              "ssecg_ce_soft_fwd_bwd")
    return dlogits, partial


def sum_partials(partial, scale=1.0, out=None):
    trace("sum_partials", tuple(getattr(partial, "shape", ())))
    partial = _req(partial, "partial")
    parts, width = partial.shape
    if out is None:
        out = torch.empty((width,), device=partial.device, dtype=torch.float32)
    check(lib().ssecg_sum_partials(_p(partial), parts, width, float(scale), _p(out), _stream()), "ssecg_sum_partials")
    return out


def loss_pair_finish(px, pu, sx_scale, su_scale):
    """-> out5 = [loss, loss, loss_x, loss_u, weight] (``ssecg_loss_pair_finish``): the two ``sum_partials`` + the torch add / mul /
    stack of the two-term losses in one launch, bit-identical."""
    trace("loss_pair_finish", tuple(px.shape), tuple(pu.shape))
    px = _req(px, "px"); pu = _req(pu, "pu")
    out = torch.empty((5,), device=px.device, dtype=torch.float32)
    check(lib().ssecg_loss_pair_finish(_p(px), px.shape[0], _p(pu), pu.shape[0], float(sx_scale), float(su_scale), _p(out), _stream()),
          "ssecg_loss_pair_finish")
    return out


# ----------------------------------------------------------------------------- optimizer
def adamw_multi(table, ntensors, max_numel, lr, beta1, beta2, eps, weight_decay, step, total_numel=0, skip_flag=None,
                skipped_count=None, refresh=None):
    """``skip_flag``: device float; non-zero -> the launch leaves everything untouched (GradScaler's inf-skip) and adds one
    to ``skipped_count`` (device float owned by the optimiser), which later launches subtract from ``step`` for the bias
    corrections."""
    weights_changed()
    trace("adamw_multi", tuple(getattr(table, "shape", ())))
    coef_dev = STEP_SCALARS.adamw_slot(lr, beta1, beta2, weight_decay, int(step), refresh) if STEP_SCALARS is not None else None
    with _Timed("adamw_multi_kernel", 0.0, 28.0 * total_numel):   # 4 reads (p, g, m, v) + 3 writes (p, m, v)
        check(lib().ssecg_adamw_multi(_p(table), ntensors, max_numel, lr, beta1, beta2, eps, weight_decay, int(step),
                                      _p(skip_flag), _p(skipped_count), coef_dev, _stream()), "ssecg_adamw_multi")


def sgd_multi(table, ntensors, max_numel, lr, momentum, weight_decay, first_step, total_numel=0, skip_flag=None, refresh=None):
    weights_changed()
    trace("sgd_multi", tuple(getattr(table, "shape", ())))
    with _Timed("sgd_multi_kernel", 0.0, (20.0 if momentum else 12.0) * total_numel):
        lr_dev = STEP_SCALARS.lr_slot(lr, refresh) if STEP_SCALARS is not None else None
        check(lib().ssecg_sgd_multi(_p(table), ntensors, max_numel, float(lr), float(momentum), float(weight_decay),
                                    int(bool(first_step)), _p(skip_flag), lr_dev, _stream()), "ssecg_sgd_multi")


def grad_norm_multi(table, ntensors, words, grad_col, numel_col, max_numel, scaler_state=None, growth_factor=2.0,
                    backoff_factor=0.5, growth_interval=2000, total_numel=0):
    """-> out (2,) f32 on the device: [global L2 norm of the gradients, found_inf]; with ``scaler_state`` (3,) f32
    = {scale, growth_tracker, skipped_steps} the GradScaler update is applied in the same launch."""
    trace("grad_norm_multi", ntensors)
    Lb = lib()
    nbytes = Lb.ssecg_grad_norm_workspace(ntensors, max_numel)
    ws = torch.empty((max(int(nbytes), 8),), device=table.device, dtype=torch.uint8)   # ~200 KB from the caching allocator
    out = torch.empty((2,), device=table.device, dtype=torch.float32)
    with _Timed("grad_sumsq_multi_kernel + grad_norm_finalize_kernel", 0.0, 4.0 * total_numel):
        check(Lb.ssecg_grad_norm_multi(_p(table), ntensors, words, grad_col, numel_col, max_numel, _p(ws), ws.numel(), _p(out),
                                       _p(scaler_state), float(growth_factor), float(backoff_factor), int(growth_interval),
                                       _stream()), "ssecg_grad_norm_multi")
    return out


def grad_clip_multi(table, ntensors, words, grad_col, numel_col, max_numel, norm, max_norm):
    trace("grad_clip_multi", ntensors)
    check(lib().ssecg_grad_clip_multi(_p(table), ntensors, words, grad_col, numel_col, max_numel, _p(norm), float(max_norm),
                                      _stream()), "ssecg_grad_clip_multi")


def ema_multi(table, ntensors, max_numel, decay):
    weights_changed()   # parameters are rewritten through raw pointers: cached Winograd operands are stale
    trace("ema_multi", tuple(getattr(table, "shape", ())))
    check(lib().ssecg_ema_multi(_p(table), ntensors, max_numel, float(decay), _stream()), "ssecg_ema_multi")


def pack_scaled_multi(table, ntensors, max_numel, dst, scale):
    """Every gradient of one reduction bucket -> its slot in the bucket's flat buffer, times ``scale`` (``ssecg_pack_scaled_multi``)."""
    trace("pack_scaled_multi", (ntensors,))
    dst = _req(dst, "dst")
    check(lib().ssecg_pack_scaled_multi(_p(table), ntensors, max_numel, _p(dst), float(scale), _stream()), "ssecg_pack_scaled_multi")


# ----------------------------------------------------------------------------- record pipeline
def strong_augment(x, plan, sigma, fs, amplitude, sine_freq, seed=0, scales=None, white=None):
    """RandAugment of B records on the device -> un-standardised (B, C, L) fp32 (``ssecg_strong_augment``)."""
    trace("strong_augment", tuple(getattr(x, "shape", ())))
    x = _req(x, "x"); plan = _req(plan, "plan", torch.int32)
    if x.dim() != 3:
        raise SsecgError("strong_augment: x must be (B, C, L)")
    B, C, L = x.shape
    Lb = lib()
    if tuple(plan.shape) != (B, 12):
        raise SsecgError("strong_augment: plan must be (B, SSECG_AUG_PLAN_WIDTH=12) int32")
    for name, t in (("scales", scales), ("white", white)):
        if t is not None and tuple(_req(t, name).shape) != (B, C, L):
            raise SsecgError(f"strong_augment: {name} must match x")
    if scales is not None: scales = _req(scales, "scales")
    if white is not None: white = _req(white, "white")
    y = torch.empty_like(x)
    check(Lb.ssecg_strong_augment(_p(x), _p(y), _p(plan), _p(scales), _p(white), B, C, L, float(sigma), float(fs),
                                  float(amplitude), float(sine_freq), int(seed) & 0xFFFFFFFFFFFFFFFF, _stream()),
          "ssecg_strong_augment")
    return y


def standardize(x, out=None):
    """Per-record (x - mean) / std over everything but the batch axis; zeros where std == 0 (``ssecg_standardize``)."""
    trace("standardize", tuple(getattr(x, "shape", ())))
    x = _req(x, "x")
    B = x.shape[0]
    y = torch.empty_like(x) if out is None else out
    check(lib().ssecg_standardize(_p(x), _p(y), B, x.numel() // B, _stream()), "ssecg_standardize")
    return y
