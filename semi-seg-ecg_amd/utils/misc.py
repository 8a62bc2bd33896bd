"""Runtime helpers with the reference's interface (``src/utils/misc.py``): meters, distributed
init over RCCL (``backend="nccl"`` on PyTorch-ROCm), the loss-scaler call convention,
checkpoint save/load, scalar all-reduce.

Differences that matter on a fast GPU: per-step statistics stay on
the device in a ``DeviceMetricBuffer`` and reach the meters in step order at print /
epoch boundaries, instead of 4 ``.item()`` + 4 scalar all-reduces + a full device
sync per step; ``math.inf`` replaces the removed ``torch._six`` (Q1).
"""
from __future__ import annotations

import builtins
import datetime
import math
import os
import time
from collections import defaultdict, deque

import torch
import torch.distributed as dist


class SmoothedValue:
    """Windowed series with median / window-avg / global-avg views (``src/utils/misc.py:14-74``)."""

    def __init__(self, window_size=20, fmt=None):
        self.deque = deque(maxlen=window_size)
        self.total = 0.0
        self.count = 0
        self.fmt = fmt or "{median:.4f} ({global_avg:.4f})"

    def update(self, value, n=1):
        self.deque.append(value)
        self.count += n
        self.total += value * n

    def synchronize_between_processes(self):
        """All-reduce count and total (the window is not synchronised)."""
        if not is_dist_avail_and_initialized():
            return
        dev = "cuda" if torch.cuda.is_available() and dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([self.count, self.total], dtype=torch.float64, device=dev)
        dist.barrier()
        dist.all_reduce(t)
        self.count, self.total = int(t[0].item()), t[1].item()

    @property
    def median(self):
        return torch.tensor(list(self.deque)).median().item()

    @property
    def avg(self):
        return torch.tensor(list(self.deque), dtype=torch.float32).mean().item()

    @property
    def global_avg(self):
        return self.total / self.count

    @property
    def max(self):
        return max(self.deque)

    @property
    def value(self):
        return self.deque[-1]

    def __str__(self):
        return self.fmt.format(median=self.median, avg=self.avg, global_avg=self.global_avg, max=self.max,
                               value=self.value)


class MetricLogger:
    def __init__(self, delimiter="\t"):
        self.meters = defaultdict(SmoothedValue)
        self.delimiter = delimiter

    def update(self, **kwargs):
        for k, v in kwargs.items():
            if v is None:
                continue
            if isinstance(v, torch.Tensor):
                v = v.item()
            assert isinstance(v, (float, int))
            self.meters[k].update(v)

    def __getattr__(self, attr):
        if attr in self.meters:
            return self.meters[attr]
        if attr in self.__dict__:
            return self.__dict__[attr]
        raise AttributeError(f"'{type(self).__name__}' object has no attribute '{attr}'")

    def __str__(self):
        return self.delimiter.join(f"{name}: {meter}" for name, meter in self.meters.items())

    def synchronize_between_processes(self):
        for meter in self.meters.values():
            meter.synchronize_between_processes()

    def add_meter(self, name, meter):
        self.meters[name] = meter

    def log_every(self, iterable, print_freq, header=None, length=None, on_print=None):
        """Yield from ``iterable``; every ``print_freq`` items print ETA / meters / timing.
        ``on_print`` (optional) runs just before a line is printed - the hook the step loops use
        to flush their device-side metric buffer into the meters."""
        header = header or ""
        n = length if length is not None else len(iterable)
        start = end = time.time()
        iter_time, data_time = SmoothedValue(fmt="{avg:.4f}"), SmoothedValue(fmt="{avg:.4f}")
        width = str(len(str(n)))
        mb = 1024.0 * 1024.0
        for i, obj in enumerate(iterable):
            data_time.update(time.time() - end)
            yield obj
            iter_time.update(time.time() - end)
            if i % print_freq == 0 or i == n - 1:
                if on_print is not None:
                    on_print()
                eta = str(datetime.timedelta(seconds=int(iter_time.global_avg * (n - i))))
                line = [header, ("[{0:" + width + "d}/{1}]").format(i, n), f"eta: {eta}", str(self),
                        f"time: {iter_time}", f"data: {data_time}"]
                if torch.cuda.is_available():
                    line.append(f"max mem: {torch.cuda.max_memory_allocated() / mb:.0f}")
                print(self.delimiter.join(line))
            end = time.time()
        total = time.time() - start
        print(f"{header} Total time: {datetime.timedelta(seconds=int(total))} ({total / max(n, 1):.4f} s / it)")


class DeviceMetricBuffer:
    """Per-step metric vectors kept on the device; ``flush`` copies the unread rows to the host once
    and feeds them to the meters in step order (same windowed medians as per-step ``.item()`` calls)."""

    def __init__(self, names, capacity, device):
        self.names = list(names)
        self.buf = torch.zeros((max(capacity, 1), len(self.names)), dtype=torch.float32, device=device)
        self.n_written = 0
        self.n_read = 0

    def push(self, values: torch.Tensor):
        """values: 1-D device tensor with one entry per name (one small device copy, no host sync)."""
        if self.n_written >= self.buf.shape[0]:
            self.buf = torch.cat((self.buf, torch.zeros_like(self.buf)))
        self.buf[self.n_written].copy_(values.detach())
        self.n_written += 1

    def flush(self, logger: MetricLogger, world_mean: bool = True):
        """-> list of per-step dicts (rank-averaged when world_size > 1) for the newly read rows.

        The all-reduce is issued by EVERY rank whenever the job is distributed - never conditioned on rank-local state
        such as "this rank has a TensorBoard writer" (only rank 0 has one; a one-sided collective would pair with another
        rank's next SyncBN / DDP collective).  The reference reduces unconditionally on every rank and gates only
        ``add_scalar`` on the writer (``src/algorithms/fixmatch.py:152-183``).  The non-finite stop is decided on the
        REDUCED values so that all ranks leave together."""
        if self.n_read == self.n_written:
            return []
        rows = self.buf[self.n_read:self.n_written]
        local = rows.cpu()
        out = []
        if world_mean and get_world_size() > 1:
            red = rows.clone()
            dist.all_reduce(red)  # ONE packed all-reduce instead of one per scalar per step
            red = (red / get_world_size()).cpu()
        else:
            red = local
        for i in range(local.shape[0]):
            vals = {k: local[i, j].item() for j, k in enumerate(self.names)}
            for j, k in enumerate(self.names):
                v = red[i, j].item()
                if not math.isfinite(v):
                    _print_all_ranks(f"Loss is {v}, stopping training")
                    raise SystemExit(1)
            logger.update(**vals)
            out.append({k: red[i, j].item() for j, k in enumerate(self.names)})
        self.n_read = self.n_written
        return out


def _print_all_ranks(msg):
    if getattr(builtins.print, "_ssecg_wrapped", False):
        print(msg, force=True)
    else:
        print(msg)


def setup_for_distributed(is_master, with_time=True):
    """Mute ``print`` on non-master ranks (``force=True`` overrides); master lines get a timestamp."""
    builtin_print = builtins.print
    if getattr(builtin_print, "_ssecg_wrapped", False):
        return

    def print(*args, **kwargs):  # noqa: A001
        force = kwargs.pop("force", False)
        if is_master or force:
            if with_time:
                builtin_print("[{}] ".format(datetime.datetime.now().time()), end="")
            builtin_print(*args, **kwargs)

    print._ssecg_wrapped = True
    builtins.print = print


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


def save_on_master(*args, **kwargs):
    if is_main_process():
        torch.save(*args, **kwargs)


def init_distributed_mode(config, with_time=True):
    """env:// rendezvous, one process per GPU (``torchrun``); ``nccl`` resolves to RCCL over xGMI on ROCm.
    ``config['dist_backend']`` may be set to ``gloo`` for CPU rehearsals (the reference forces nccl)."""
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        config["rank"] = int(os.environ["RANK"])
        config["world_size"] = int(os.environ["WORLD_SIZE"])
        config["gpu"] = int(os.environ.get("LOCAL_RANK", 0))
    elif "SLURM_PROCID" in os.environ:
        config["rank"] = int(os.environ["SLURM_PROCID"])
        config["gpu"] = config["rank"] % max(torch.cuda.device_count(), 1)
    else:
        print("Not using distributed mode")
        setup_for_distributed(is_master=True, with_time=with_time)
        config["distributed"] = False
        return
    config["distributed"] = True
    backend = config.get("dist_backend") or "nccl"
    if backend == "nccl":
        torch.cuda.set_device(config["gpu"])
    config["dist_backend"] = backend
    print(f"| distributed init (rank {config['rank']}): {config.get('dist_url', 'env://')}, gpu {config['gpu']}", flush=True)
    if not dist.is_initialized():
        # SSECG_DIST_TIMEOUT_S (default 600): a rank stuck in a collective longer than this aborts (the backend's watchdog)
        # and torchrun takes the other ranks down with it, instead of a silent hang (the reference sets no timeout:
        # src/utils/misc.py:226-229 -> torch's 10 minutes for NCCL)
        dist.init_process_group(backend=backend, init_method=config.get("dist_url", "env://"),
                                world_size=config["world_size"], rank=config["rank"],
                                timeout=datetime.timedelta(seconds=float(os.environ.get("SSECG_DIST_TIMEOUT_S", "600"))))
    dist.barrier()
    setup_for_distributed(config["rank"] == 0, with_time=with_time)


class NativeScalerWithGradNormCount:
    """Call convention and observable state of ``src/utils/misc.py:236-263`` (``torch.cuda.amp.GradScaler`` with its
    defaults, live on a GPU even when ``use_amp`` is false).

    fp32 hot path: multiplying the loss by ``scale`` (a power of two) and dividing the gradients by it again is exact in
    fp32 outside the overflow / subnormal ranges, so the loss is back-propagated unscaled; everything else GradScaler does
    is reproduced ON THE DEVICE, without a host round trip per step: one launch pair computes the global gradient norm
    (the value ``get_grad_norm_`` returns), ``found_inf`` and ``GradScaler.update()`` - ``scale`` halves and the step is
    skipped inside the optimiser kernel when a gradient is inf/NaN, ``scale`` doubles after ``growth_interval`` (2000)
    clean steps - so the checkpoint's ``scaler`` entry moves exactly like the reference's."""
    state_dict_key = "amp_scaler"

    def __init__(self):
        self._host = {"scale": 65536.0, "growth_factor": 2.0, "backoff_factor": 0.5, "growth_interval": 2000,
                      "_growth_tracker": 0}
        self._dev = None   # float32 [scale, growth_tracker, skipped_steps] on the gradients' device

    def _device_state(self, device):
        if self._dev is None or self._dev.device != device:
            self._dev = torch.tensor([self._host["scale"], float(self._host["_growth_tracker"]), 0.0], dtype=torch.float32,
                                     device=device)
        return self._dev

    def __call__(self, loss, optimizer, clip_grad=None, parameters=None, create_graph=False, update_grad=True):
        from ssecg import ops
        if ops.STEP_SCALARS is not None:
            # the step is being captured into a HIP graph (ssecg/graph.py): the backward below only RECORDS launches, and
            # AccumulateGrad points every ``.grad`` at memory of the capture pool that no kernel has written.  If the capture
            # fails, the eager re-run must not accumulate into that memory: drop the gradients in the abort path.
            plist = [p for g in optimizer.param_groups for p in g["params"]]
            ops.STEP_SCALARS.aborts.append(lambda plist=plist: [setattr(p, "grad", None) for p in plist])
        loss.backward(create_graph=create_graph)
        if not update_grad:
            return None
        if getattr(optimizer, "supports_found_inf", False):
            dev = next((p.device for g in optimizer.param_groups for p in g["params"] if p.grad is not None), None)
            if dev is None:
                return torch.tensor(0.0)
            h = self._host
            out = optimizer.grad_norm(self._device_state(dev), h["growth_factor"], h["backoff_factor"], h["growth_interval"],
                                      max_norm=clip_grad)
            optimizer.step(found_inf=out[1:2])
            return out[0]
        # an optimiser that is not one of the fused HIP ones (rehearsals): plain path, scaler state left alone
        if clip_grad is not None:
            assert parameters is not None
            norm = torch.nn.utils.clip_grad_norm_(parameters, clip_grad)
        else:
            norm = get_grad_norm_(parameters)
        optimizer.step()
        return norm

    def state_dict(self):
        if self._dev is not None:
            v = self._dev.tolist()   # one device read, at checkpoint time only
            self._host["scale"], self._host["_growth_tracker"] = float(v[0]), int(v[1])
        return dict(self._host)

    def load_state_dict(self, state_dict):
        if state_dict:
            self._host.update(state_dict)
            self._dev = None


def get_grad_norm_(parameters, norm_type: float = 2.0) -> torch.Tensor:
    if isinstance(parameters, torch.Tensor):
        parameters = [parameters]
    grads = [p.grad.detach() for p in parameters if p.grad is not None] if parameters is not None else []
    if not grads:
        return torch.tensor(0.0)
    if float(norm_type) == math.inf:
        return torch.stack([g.abs().max() for g in grads]).max()
    return torch.linalg.vector_norm(torch.stack(torch._foreach_norm(grads, norm_type)), norm_type)


def save_model(config, checkpoint_path, epoch, model_without_ddp, optimizer=None, loss_scaler=None, metrics=None,
               model_ema=None):
    """Checkpoint schema of ``src/utils/misc.py:281-302``: epoch, model, optimizer, scaler, config[, metrics][, model_ema]."""
    to_save = {
        "epoch": epoch,
        "model": model_without_ddp.state_dict(),
        "optimizer": optimizer.state_dict() if optimizer is not None else None,
        "scaler": loss_scaler.state_dict() if loss_scaler is not None else None,
        "config": config,
    }
    if metrics is not None:
        to_save["metrics"] = metrics
    if model_ema is not None:
        to_save["model_ema"] = model_ema.state_dict()
    save_on_master(to_save, checkpoint_path)


def load_model(config, model_without_ddp, optimizer, loss_scaler, model_ema=None):
    if not config.get("resume"):
        return
    if str(config["resume"]).startswith("https"):
        checkpoint = torch.hub.load_state_dict_from_url(config["resume"], map_location="cpu", check_hash=True)
    else:
        checkpoint = torch.load(config["resume"], map_location="cpu", weights_only=False)
    model_without_ddp.load_state_dict(checkpoint["model"])
    if model_ema is not None and "model_ema" in checkpoint:
        ema_sd = checkpoint["model_ema"]
        for name, buf in model_ema.named_buffers():  # float32 counters written by the EMA (Q5)
            if name in ema_sd and ema_sd[name].dtype != buf.dtype:
                buf.data = buf.data.to(ema_sd[name].dtype)
        model_ema.load_state_dict(ema_sd)
    print("Resume checkpoint %s" % config["resume"])
    if "optimizer" in checkpoint and "epoch" in checkpoint and not config.get("eval"):
        optimizer.load_state_dict(checkpoint["optimizer"])
        config["start_epoch"] = checkpoint["epoch"] + 1
        if checkpoint.get("scaler") is not None:
            loss_scaler.load_state_dict(checkpoint["scaler"])
        print("With optim & sched!")


def all_reduce_mean(x):
    world_size = get_world_size()
    if world_size == 1:
        return x
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor(x, dtype=torch.float32, device=dev)
    dist.all_reduce(t)
    return (t / world_size).item()


@torch.no_grad()
def concat_all_gather(tensor):
    world_size = get_world_size()
    if world_size == 1:
        return tensor
    gathered = [torch.empty_like(tensor) for _ in range(world_size)]
    dist.all_gather(gathered, tensor, async_op=False)
    return torch.cat(gathered, dim=0)
