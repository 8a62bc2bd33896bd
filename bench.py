#!/usr/bin/env python3
"""Headline benchmark: ECG windows/s of the FixMatch training step on MI355X.

Workload (BASELINE.json metric / SURVEY.md §8d headline config): FixMatch, ResNet18-1D + FCNHead,
B = 512 labelled + 512 unlabelled windows per GPU per step (weak + strong views), 12 leads, L = 2000,
fp32, AdamW, SyncBN + DDP when N > 1.  A "step" = teacher pass (eval) + student pass over 2B windows +
both losses + backward + AdamW (reference: src/algorithms/fixmatch.py:79-140); inputs are synthetic and
resident in HBM before the timed region.

  python bench.py --gpus N --steps K --warmup W

N > 1 without WORLD_SIZE in the environment: this process starts ``python -m torch.distributed.run
--nnodes=1 --nproc-per-node N ... bench.py`` as a CHILD (before anything here touches the GPU; never an
exec), relays the child's JSON line and exits with its code - the launch ``scripts/train.sh:108-141`` of
the reference does with torchrun.  With WORLD_SIZE set (the driver's own torchrun launch) it is a rank.

Prints ONE JSON line on rank 0 (contract in the task statement) including
  "roofline"      : the dominant kernel (HIP events on the launch stream during one extra instrumented step outside the
                    timed region): ``achieved`` / ``frac`` = the multiplications the kernel EXECUTES over its time against
                    the MFMA peak (Winograd F(4,3) issues 1/2, F(2,3) 2/3 of the direct convolution's; never above 1),
                    ``achieved_algorithmic`` = the direct-convolution FLOPs of SURVEY 8d over the same time (TFLOP/s);
                    ``ceiling_model`` = what the kernel's instruction mix and held clock allow (fp32 MFMAs share the FMA lanes
                    with vector instructions: DESIGN.md section 5) - ``frac / attainable_frac`` is the share of THAT it reaches,
  "kernel_classes": every kernel class of the step with its own roof (max of FLOPs / 157.3 TF and bytes / 8 TB/s),
  "cpu_baseline"  : the oracle (oracle/torch_ref.py, a CPU restatement) timed on this box's host cores, B = 16 and
                    B = 64, 3 warm-up + 10 timed steps each, rank 0 at N = 1 only.
  "amp"           : a SECOND timed region in the same process, after the headline one, same protocol (W warm-up + K timed
                    steps + one instrumented step): the reduced-precision student pass the reference's default
                    ``use_amp: true`` selects (bf16 storage + bf16 MFMA) - ms_per_step, windows_per_s,
                    roofline, kernel_classes, step_roofline of that region.  The headline metric / dtype / value are
                    untouched (fp32).  ``--no-amp-record`` skips it.
``--amp`` makes the reduced-precision path the headline region instead (dtype "bf16"; no sub-record).
"""
import argparse
import gc
import glob
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(ROOT, "semi-seg-ecg_amd")
for p in (ROOT, SRC):
    if p not in sys.path:
        sys.path.insert(0, p)

PEAK_FP32_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 matrix == vector peak
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA
PEAK_HBM_TBS = 8.0
MAC_BASE, MAC_PER_LEAD = 353_828_352, 448_000   # conv MACs per window, forward, L=2000
# algorithmic HBM bytes per unit of B per FixMatch step, "materialise-once" model of SURVEY.md §8d: W elements per window
# (conv outputs 681,596 + BN outputs 681,344 + max-pool 32,000 + dropout 8,064 + up-sampled logits 8,000), train forward =
# write + read once (2 W), backward = read saved + write grad + read grad (3 W); eval forward (BN folded, ReLU fused) 5.77 MB
W_ELEMS = 1_411_004
W_FP32_UNDER_AMP = 64_000 + 64_000 + 32_000 + 8_064 + 8_000   # stem conv / BN / pool outputs, dropout, logits stay 4 B
EVAL_BYTES = 2 * (681_596 + 32_000 + 8_000) * 4                 # teacher pass: fp32 on both lines (outside autocast)
BYTES_PER_B = EVAL_BYTES + 2 * 5 * W_ELEMS * 4                  # = 62.2e6 (fp32 storage)
BYTES_PER_B_AMP = EVAL_BYTES + 2 * 5 * (W_FP32_UNDER_AMP * 4 + (W_ELEMS - W_FP32_UNDER_AMP) * 2)   # = 37.5e6
PARAM_BYTES = 0.2e9


def model_config(C):
    return {"backbone": {"resnet18": dict(num_leads=C, num_stages=4, out_indices=[0, 1, 2, 3], dilations=[1, 1, 1, 1],
                                           strides=[1, 2, 2, 2], deep_stem=False, avg_down=False, contract_dilation=False)},
            "decode_head": {"FCNHead": dict(in_channels=512, in_index=3, channels=128, num_convs=1, concat_input=False,
                                            dropout_ratio=0.1, num_classes=4, align_corners=False)}}


TRAIN_CFG = dict(epochs=100, accum_iter=1, warmup_epochs=10, min_lr=1e-4, lr=1e-3, weight_decay=0.05, max_norm=None,
                 optimizer="adamw", optimizer_kwargs={"betas": [0.9, 0.999]}, conf_thresh=0.80)


def kernel_source_hash():
    """sha256 over the kernel sources + the C-ABI header: ties a committed PMC traffic summary to the code it measured."""
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(SRC, "csrc", "*.hip")) + glob.glob(os.path.join(SRC, "csrc", "*.h")))
    for f in files + [os.path.join(ROOT, "include", "ssecg.h")]:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


_BATCH_CACHE = {}


def synthetic_batch(B, C, L, seed, device):
    """The synthetic FixMatch batch of SURVEY.md §8d - ecg ~ N(0,1), strong = weak + 0.5 N(0,1), piecewise-constant labels - from
    the repo's own counter generator (``ssecg.synth``: splitmix64 -> Box-Muller, pure numpy) with seed 1234 + rank, the generator
    the CPU baseline and the parity tests draw from: the line's ``final_stats`` are regenerable off the box (round-5 review:
    ``torch.randn`` on the device was not)."""
    import torch
    from ssecg import synth
    key = (seed, B, C, L)
    if key not in _BATCH_CACHE:          # (the fp32 and the bf16 timed regions read the same batch: ~8 s of numpy at 512 x 12 x 2000)
        _BATCH_CACHE.clear()
        _BATCH_CACHE[key] = synth.fixmatch_batch(seed, B, C, L)
    b = _BATCH_CACHE[key]
    t = lambda a: torch.from_numpy(a).to(device)
    return t(b["labeled"]["ecg"]), t(b["labeled"]["target"]), t(b["unlabeled"]["ecg"]), t(b["unlabeled"]["ecg_aug"])


def cpu_baseline(C, L, batches=(16, 64), warm=3, steps=10, budget_s=12.0):
    """The oracle's FixMatch step on the host cores (rank 0, N = 1 only): a BOUNDED sample of the same workload,
    SURVEY.md §8d protocol (B = 16 and 64, 3 warm-up + up to 10 timed steps, all host cores).  The hosts of the pool differ by 4 x in
    what they leave to this leg (1.5-6.8 s per B = 64 step, measured in round 6): the timed steps of a batch size stop after
    ``budget_s`` seconds (never fewer than 3), so the default bench run finishes within a few minutes on every box."""
    import torch
    from oracle import torch_ref as O
    from ssecg import synth
    cfg = dict(TRAIN_CFG); cfg["betas"] = (0.9, 0.999)
    runs = []
    for Bc in batches:
        sd = O.state_from_numpy(synth.model_state(0, C))
        batch = {g: {k: torch.from_numpy(v) for k, v in d.items()} for g, d in synth.fixmatch_batch(1234, Bc, C, L).items()}
        opt = {}
        tw = time.time()
        nwarm = 0
        for _ in range(warm):
            O.fixmatch_step(sd, opt, batch, cfg, 3.0)
            nwarm += 1
            if time.time() - tw > budget_s / 2 and nwarm >= 1:
                break
        t0 = time.time()
        done = 0
        for _ in range(steps):
            O.fixmatch_step(sd, opt, batch, cfg, 3.0)
            done += 1
            if done >= 3 and time.time() - t0 > budget_s:
                break
        dt = (time.time() - t0) / done
        runs.append({"B": Bc, "ms_per_step": dt * 1e3, "windows_per_s": Bc / dt, "timed_steps": done, "warmup_steps": nwarm})
    best = max(runs, key=lambda r: r["windows_per_s"])
    return {"value": best["windows_per_s"], "unit": "windows/s", "cores": torch.get_num_threads(), "cpu": cpu_model(),
            "kind": "port", "runs": runs,
            "sample": f"oracle/torch_ref.fixmatch_step, C={C}, L={L}, fp32, B in {list(batches)}, up to {warm} warm-up + {steps} timed "
                      f"steps each, bounded to ~{budget_s:.0f} s per batch size; value = the faster of the two (B={best['B']}, "
                      f"{best['ms_per_step']:.0f} ms/step over {best['timed_steps']} steps)"}


def wino_executed(name):
    """Multiplications a kernel's algorithm issues per direct-convolution multiplication: F(4,3) and its transpose (the weight
    gradient conv_wino_wgrad4_kernel) 6/12, F(2,3) and its transpose 8/12, direct kernels 1."""
    if "wino" not in name:
        return 1.0
    return 0.5 if ("wino4" in name or "wgrad4" in name) else 2.0 / 3.0


def kernel_class(name):
    if name.startswith("conv_wino_wgrad"):
        return "winograd_wgrad"
    if name.startswith("conv_wino"):
        return "winograd_fwd_dgrad"
    if name.startswith("conv_wgrad_b16"):
        return "bf16_wgrad"
    if name.startswith("conv_b16"):
        return "bf16_fwd_dgrad"
    if name.startswith("stem_"):
        return "stem"
    if name.startswith("conv_igemm_fast"):
        return "direct_fwd_dgrad"
    if name.startswith("conv_igemm_kernel"):
        return "generic_igemm"
    if name.startswith("conv_wgrad_kernel"):
        return "direct_wgrad"
    if name.startswith("bn_"):
        return "batchnorm_elementwise"
    return "other_elementwise"


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def launch_ranks(args, argv):
    """Parent of an N-rank run: one child torchrun (NOT an exec, nothing here has touched the GPU), JSON line relayed.
    A rank stuck in a collective makes no progress and prints nothing: when the child has been silent for
    SSECG_DIST_TIMEOUT_S (default 300 s; the ranks' process group uses the same value as its collective timeout) plus a
    grace minute, the parent terminates the CHILD process group it started and exits 124 - nothing is re-executed."""
    import signal
    import threading
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    limit = float(env.get("SSECG_DIST_TIMEOUT_S", "300")) + 60.0
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)
    state = {"line": None, "last": time.monotonic()}

    def reader():
        for out in proc.stdout:
            state["last"] = time.monotonic()
            out = out.rstrip("\n")
            if out.startswith("{") and '"metric"' in out:
                state["line"] = out
            else:
                print(out, file=sys.stderr)

    th = threading.Thread(target=reader, daemon=True)
    th.start()
    rc = None
    while rc is None:
        try:
            rc = proc.wait(timeout=5.0)
        except subprocess.TimeoutExpired:
            if time.monotonic() - state["last"] > limit:
                print(f"bench.py: the ranks have been silent for {limit:.0f} s (a hung collective?): terminating them", file=sys.stderr)
                for sig in (signal.SIGTERM, signal.SIGKILL):
                    try:
                        os.killpg(proc.pid, sig)          # the session this parent created for its own child, nothing else
                    except ProcessLookupError:
                        break
                    try:
                        proc.wait(timeout=15.0)
                        break
                    except subprocess.TimeoutExpired:
                        continue
                sys.exit(124)
    th.join(5.0)
    if state["line"] is not None:
        print(state["line"], flush=True)
    elif rc == 0:
        rc = 1
        print("bench.py: the ranks exited without printing a result line", file=sys.stderr)
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=512, help="labelled (= unlabelled) windows per GPU per step")
    ap.add_argument("--leads", type=int, default=12)
    ap.add_argument("--length", type=int, default=2000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--amp", action="store_true", help="reduced-precision path (use_amp: true): bf16 storage + bf16 MFMA")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL over xGMI on ROCm); gloo only for rehearsals")
    ap.add_argument("--no-amp-record", action="store_true", help="skip the second timed region (the bf16 sub-record `amp`)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the whole step as ONE HIP graph after two eager steps (ssecg/graph.py); under torch.distributed "
                         "over RCCL the collectives are captured with the step")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args, sys.argv[1:])   # does not return

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    share = os.environ.get("SSECG_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0  # rehearsal only: several ranks on one card (use --backend gloo; RCCL needs one GPU per rank)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # SSECG_BENCH_FORCE_DIST=1 (rehearsal on a one-GPU box): a world-size-1 RCCL group, DDP wrap and forced SyncBN all-reduces -
    # every collective of the N > 1 step is issued through ProcessGroupNCCL (the line says so in config.parallelism)
    force_dist = world == 1 and os.environ.get("SSECG_BENCH_FORCE_DIST") == "1"
    distributed = world > 1 or force_dist
    if force_dist:
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        os.environ["SSECG_FORCE_SYNC_COLLECTIVES"] = "1"
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        # a rank stuck in a collective longer than this aborts (ProcessGroupNCCL's watchdog) instead of hanging the job; the
        # launcher parent enforces the same limit from outside (launch_ranks)
        dist.init_process_group(backend=args.backend, init_method="env://", world_size=world, rank=rank,
                                timeout=datetime.timedelta(seconds=float(os.environ.get("SSECG_DIST_TIMEOUT_S", "300"))))
        assert dist.get_world_size() == args.gpus and dist.get_backend() == args.backend
        if args.backend == "nccl":
            # every rank must have reached RCCL on its own GPU: one all-reduce of ones counts the ranks
            probe = torch.ones(1, device=device)
            dist.all_reduce(probe)
            assert int(probe.item()) == args.gpus, f"RCCL saw {int(probe.item())} ranks, expected {args.gpus}"

    import utils.lr_sched as lr_sched
    from algorithms.base import init_model_from_cfg, wrap_ddp
    from algorithms.fixmatch import fixmatch_step
    from ssecg import ops
    from utils.misc import DeviceMetricBuffer, NativeScalerWithGradNormCount
    from utils.optimizer import get_optimizer_from_config

    B, C, L = args.batch, args.leads, args.length

    def measure(amp):
        """One timed region (W warm-up + K timed steps + one instrumented step) of the FixMatch step; amp = the reduced-
        precision student pass (use_amp: true).  -> the bench record of that region (rank 0) or None."""
        torch.manual_seed(0)  # identical random-init weights on every rank (reference init law)
        model = init_model_from_cfg(model_config(C)).to(device)
        if amp:
            from ssecg import amp as SAMP
            SAMP.enable(model)
        # SyncBN conversion + DDP exactly as the plugins' train(config) does it (algorithms/base.py:wrap_ddp)
        model, model_without_ddp = wrap_ddp({"ddp": {"distributed": distributed, "sync_bn": True, "gpu": local_rank}}, model)
        cfg = dict(TRAIN_CFG)
        optimizer = get_optimizer_from_config(cfg, model_without_ddp.parameters())
        scaler = NativeScalerWithGradNormCount()
        ecg_x, mask_x, ecg_u_w, ecg_u_s = synthetic_batch(B, C, L, 1234 + rank, device)
        total = args.warmup + args.steps + 1
        buf = DeviceMetricBuffer(['loss_total', 'loss_x', 'loss_u_s', 'mask_ratio'], total, device)

        def whole_step(ecg_x, mask_x, ecg_u_w, ecg_u_s):
            loss, stats = fixmatch_step(model, ecg_x, mask_x, ecg_u_w, ecg_u_s, cfg['conf_thresh'])
            scaler(loss, optimizer, clip_grad=None, parameters=model.parameters(), update_grad=True)
            optimizer.zero_grad()
            return stats

        graphed = None
        if args.graph:
            if distributed and args.backend != "nccl":
                raise SystemExit("bench.py --graph under torch.distributed: RCCL (--backend nccl) only - gloo's collectives are staged "
                                 "through the host and cannot be captured")
            from ssecg.graph import StepGraph
            graphed = StepGraph(whole_step)

        def one_step(i, eager=False):
            lr_sched.adjust_learning_rate(optimizer, 10.0 + i / 1000.0, cfg)
            buf.push((whole_step if (graphed is None or eager) else graphed)(ecg_x, mask_x, ecg_u_w, ecg_u_s))

        for i in range(args.warmup):
            one_step(i)
        if graphed is not None and graphed.graph is None:
            raise SystemExit("bench.py --graph: --warmup must be at least 3 (two eager steps, then the capture)")
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for i in range(args.steps):
            one_step(args.warmup + i)
        ev1.record()
        host_issue = time.perf_counter() - t0      # host time spent ISSUING the K steps (before anything waits for the device)
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        dev_ms = ev0.elapsed_time(ev1)
        t = torch.tensor([wall], dtype=torch.float64, device=device)
        rank_walls = [wall]
        if distributed:
            gathered = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(gathered, t)
            rank_walls = [g.item() for g in gathered]
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = t.item()

        # ---- one extra instrumented step: HIP events around every launch on the launch stream ----
        from ssecg import functional as SF
        ops.PROFILE = []
        SF.COLLECTIVE_LOG = [] if distributed else None
        one_step(total - 1, eager=True)
        torch.cuda.synchronize()
        prof, ops.PROFILE = ops.PROFILE, None
        coll, SF.COLLECTIVE_LOG = SF.COLLECTIVE_LOG, None
        per = {}
        for name, flops, e0, e1, nbytes in prof:
            d = per.setdefault(name, [0.0, 0.0, 0, 0.0])
            d[0] += flops; d[1] += e0.elapsed_time(e1) * 1e-3; d[2] += 1; d[3] += nbytes
        conv = {k: v for k, v in per.items() if v[0] > 0}
        dom = max(conv.items(), key=lambda kv: kv[1][1])
        conv_time = sum(v[1] for v in conv.values())

        if rank == 0:
            peak_mm = PEAK_BF16_TFLOPS if amp else PEAK_FP32_TFLOPS
            dtype = "bf16" if amp else "f32"
            ms_per_step = wall / args.steps * 1e3
            value = world * B * args.steps / wall
            mac = MAC_BASE + MAC_PER_LEAD * C if L == 2000 else None
            out = {
                "metric": f"ECG windows/sec (FixMatch step, ResNet18-1D + FCNHead, B={B}/GPU, L={L}, {C}-lead, {dtype}; whole job = "
                          f"per-GPU x n_gpus)",
                "value": value, "unit": "windows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": dtype, "data": "synthetic",
                "config": {"workload": f"FixMatch step, {B} labelled + {B} unlabelled windows/GPU (weak+strong views), "
                                       f"{C} leads, L={L}, ResNet18-1D + FCNHead, AdamW, random-init weights",
                           "global_batch": world * B, "parallelism": f"dp{world}" + ("+syncbn" if distributed else "") +
                                                                      (" (one-rank RCCL rehearsal: collectives forced)" if force_dist else ""),
                           "backend": (dist.get_backend() if distributed else None),
                           "ranks_share_one_gpu": bool(share) if distributed else False,
                           "hip_graph": (f"whole step replayed as one HIP graph ({graphed.replays} replays)" if graphed is not None else False)},
                "per_gpu_windows_per_s": value / world,
                "device_ms_per_step": dev_ms / args.steps,
                # host time spent issuing one step (launch plumbing + autograd + optimiser bookkeeping; rank 0): host-bound when it
                # approaches ms_per_step - visible as the bf16 / small-batch lines get faster
                "host_ms_per_step": host_issue / args.steps * 1e3,
            }
            from ssecg import config as _switches
            out["config"]["switches"] = _switches.non_default()     # every SSECG_* switch off its default: the line reproduces itself
            if distributed:
                # what a first N > 1 run needs to be diagnosable from its one line: the collectives ONE step issues (the same
                # sequence on every rank - asserted by the 2/4/8-rank tests) and how far the ranks' own clocks are apart
                kinds = {}
                for kind, numel, dt in coll:
                    k = kinds.setdefault(kind, {"count": 0, "elements": 0, "dtype": dt})
                    k["count"] += 1; k["elements"] += int(numel)
                out["dist"] = {"collectives_per_step": kinds,
                               "rank_ms_per_step": {"min": min(rank_walls) / args.steps * 1e3, "max": max(rank_walls) / args.steps * 1e3,
                                                    "per_rank": [w / args.steps * 1e3 for w in rank_walls]},
                               "collective_timeout_s": float(os.environ.get("SSECG_DIST_TIMEOUT_S", "300"))}
            dname, (dfl, dsec, dn, dby) = dom

            def kernel_peak(name):   # the matrix pipe a kernel runs on: the bf16 kernels of csrc/amp.hip carry "b16" in their names
                return PEAK_BF16_TFLOPS if "b16" in name else PEAK_FP32_TFLOPS

            dpeak = kernel_peak(dname)
            ach = dfl / dsec / 1e12
            is_wino = "wino" in dname
            # multiplications the kernel's algorithm issues per direct-conv multiplication: F(4,3) 6/12, F(2,3) 8/12
            wino_exec = wino_executed(dname)
            hbm_bound = dby and (dby / (PEAK_HBM_TBS * 1e12) > dfl / (dpeak * 1e12))
            if hbm_bound:
                gbs = dby / dsec / 1e9
                out["roofline"] = {"bound": "hbm", "kernel": dname, "achieved": gbs, "peak": PEAK_HBM_TBS * 1e3, "unit": "GB/s",
                                   "frac": gbs / (PEAK_HBM_TBS * 1e3), "tflops": ach}
            else:
                # achieved / frac count the multiplications the matrix pipe EXECUTES (a Winograd kernel issues 1/2 or 2/3 of the
                # direct convolution's): a true fraction of the MFMA peak, never above 1.  The direct-conv-equivalent rate is
                # reported beside it in TFLOP/s (achieved_algorithmic), not as a fraction.
                out["roofline"] = {"bound": "mfma", "kernel": dname, "achieved": ach * wino_exec, "peak": dpeak, "unit": "TFLOP/s",
                                   "frac": ach * wino_exec / dpeak, "achieved_algorithmic": ach,
                                   "executed_over_algorithmic_multiplications": wino_exec}
            out["roofline"].update({"traffic": None, "launches_per_step": dn, "avg_launch_ms": dsec / dn * 1e3,
                                    "algorithmic_flops_per_launch": dfl / dn,
                                    "algorithmic_bytes_per_launch": (dby / dn) if dby else None,
                                    "conv_ms_per_step": conv_time * 1e3})
            if is_wino:
                out["roofline"]["note"] = ("Winograd " + ("F(4,3): 6 instead of 12" if wino_exec == 0.5 else "F(2,3): 8 instead of 12") +
                                           " multiplications per four outputs and channel pair.  achieved / frac = executed "
                                           "multiplications over kernel time against the fp32 MFMA peak; achieved_algorithmic = the "
                                           "direct-convolution FLOPs (SURVEY 8d) over the same time, in TFLOP/s")
            # ---- per-kernel-class table: each class against ITS OWN roof ----
            classes = {}
            for name, (fl, sec, n, by) in per.items():
                c = classes.setdefault(kernel_class(name), {"ms_per_step": 0.0, "launches": 0, "flops": 0.0, "bytes": 0.0, "kernels": {}})
                c["ms_per_step"] += sec * 1e3; c["launches"] += n; c["flops"] += fl; c["bytes"] += by
                c["kernels"][name] = {"ms_per_step": sec * 1e3, "launches": n,
                                      **({"tflops": fl / sec / 1e12} if fl else {"gb_per_s": by / sec / 1e9})}
            for name, (fl, sec, n, by) in per.items():   # compute time of a class at each kernel's own matrix-pipe peak
                c = classes[kernel_class(name)]
                c["_tc"] = c.get("_tc", 0.0) + fl / (kernel_peak(name) * 1e12)
            for cname, c in classes.items():
                t_c, t_m = c.pop("_tc", 0.0), c["bytes"] / (PEAK_HBM_TBS * 1e12)
                c["bound"] = "mfma" if t_c >= t_m else "hbm"
                if cname.startswith("winograd"):
                    # the roof of a Winograd class is the time of the multiplications it EXECUTES at the MFMA peak (1/2 or 2/3 of
                    # the direct convolution's, per kernel) - a fraction of its own roof is then never above 1
                    t_c = sum(per[k][0] * wino_executed(k) / (kernel_peak(k) * 1e12) for k in c["kernels"])
                    c["algorithmic_tflops"] = c["flops"] / (c["ms_per_step"] * 1e-3) / 1e12 if c["ms_per_step"] else None
                c["roof_ms"] = max(t_c, t_m) * 1e3
                c["frac_of_own_roof"] = c["roof_ms"] / c["ms_per_step"] if c["ms_per_step"] else None
            out["kernel_classes"] = dict(sorted(classes.items(), key=lambda kv: -kv[1]["ms_per_step"]))
            # measured HBM traffic: PMC passes cannot run inside this process, so the per-launch figure comes from the committed
            # rocprofv3 summary of this same command (profiles/README.md) - only if it was taken on THESE kernel sources and
            # THIS workload; otherwise null with the reason
            tr = {"traffic": None}
            try:
                files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")), key=os.path.getmtime)
                cands = []
                for f in files:
                    j = json.load(open(f))
                    cands.append((f, j))
                want = {"B": B, "C": C, "L": L, "dtype": dtype}
                cur = kernel_source_hash()
                match = [(f, j) for f, j in cands if j.get("source_hash") == cur and j.get("workload") == want]
                if match:
                    tfile, tj = match[-1]
                    allk = tj["kernels"]
                    key = dname.split(" (")[0] if " (" in dname else dname
                    tk = allk.get(key)
                    if tk is None:
                        # rocprofv3 prints every template argument (conv_wino4_kernel<4, 2, false> / <4, 2, true>); the timer name
                        # stops at the tile shape: launch-weighted mean over the instances
                        inst = [v for k, v in allk.items() if k.startswith(key.rstrip(">"))]
                        nl = sum(v["launches_per_step"] for v in inst)
                        if inst and nl > 0:
                            tk = {"read_bytes": sum(v["read_bytes"] * v["launches_per_step"] for v in inst) / nl,
                                  "write_bytes": sum(v["write_bytes"] * v["launches_per_step"] for v in inst) / nl}
                    if tk:
                        tr = {"traffic": tk["read_bytes"] + tk["write_bytes"],
                              "traffic_detail": {"read_bytes": tk["read_bytes"], "write_bytes": tk["write_bytes"],
                                                 "source": os.path.relpath(tfile, ROOT), "source_hash": cur}}
                        # SQ counters of the same build (tools/summarize_pmc.py): matrix-pipe busy share, waiting waves, LDS
                        pm = [(v["pmc"], v["launches_per_step"]) for k, v in allk.items()
                              if k.startswith(key.rstrip(">")) and "pmc" in v]
                        nl = sum(n for _, n in pm)
                        if pm and nl > 0:
                            for fld, name in (("mfma_busy", "mfma_busy"), ("waiting", "waves_waiting"), ("lds_conflict", "lds_bank_conflicts")):
                                vals = [(q[fld], n) for q, n in pm if fld in q]
                                if vals:
                                    tr[name] = sum(v * n for v, n in vals) / sum(n for _, n in vals)
                            tr["pmc_source"] = tj.get("pmc_source")
                            # Ceiling model of an fp32 MFMA kernel (measured: profiles/r06_mfma_valu_mix.txt,
                            # r06_pmc_valu_per_mfma.md): fp32 MFMAs and vector instructions share the FMA lanes (co-execution cycles
                            # 0), so with v vector instructions beside each MFMA the pipe holds at most 64 / (68 + 2 v) at two waves
                            # per SIMD, times the clock the kernel holds under load (1.9 of 2.4 GHz: profiles/r02_ablation_wino4.txt).
                            vv = [(q["valu_beside_mfma"], n) for q, n in pm if "valu_beside_mfma" in q]
                            if vv and not amp:
                                v = sum(x * n for x, n in vv) / sum(n for _, n in vv)
                                cyc = 64.0 + (4.0 + 2.0 * v if v > 0.5 else 6.0 * v)
                                held = 1.9
                                tr["ceiling_model"] = {
                                    "vector_insts_beside_each_mfma": v, "mfma_valu_coexec_cycles": max(q.get("mfma_valu_coexec_cycles", 0.0) for q, _ in pm),
                                    "cycles_per_mfma": cyc, "issue_frac": 64.0 / cyc, "held_clock_ghz": held, "clock_frac": held / 2.4,
                                    "attainable_frac": 64.0 / cyc * held / 2.4,
                                    "note": "frac / attainable_frac = how much of what the instruction mix and the held clock allow the kernel "
                                            "reaches; the rest are barrier / wait stalls (waves_waiting)",
                                    "source": tj.get("pmc_valu_source")}
                    hb = sum((v["read_bytes"] + v["write_bytes"]) * v["launches_per_step"] for v in allk.values())
                    tr["measured_hbm_bytes_per_step"] = hb
                else:
                    tr["traffic_reason"] = (f"no committed profiles/*_traffic.json matches kernel source hash {cur} and workload {want} "
                                            "(PMC summary is stale for this build: re-run tools/profile_bench.sh)")
            except Exception as e:  # noqa: BLE001
                tr["traffic_reason"] = f"could not read profiles/*_traffic.json: {e}"
            mb = tr.pop("measured_hbm_bytes_per_step", None)
            out["roofline"].update(tr)
            if mac is not None:
                # F = 14*B*MAC FLOPs (2 FLOP/MAC x [teacher B + student 2B] forward + 4 FLOP/MAC x 2B backward)
                F = 14.0 * B * mac
                A = (BYTES_PER_B_AMP if amp else BYTES_PER_B) * B + PARAM_BYTES   # --amp: teacher pass + stem + head tail at 4 B
                # use_amp: the student's 12*B*MAC run on the bf16 pipe, the teacher forward (outside autocast) stays fp32
                t_c = (12.0 * B * mac / (PEAK_BF16_TFLOPS * 1e12) + 2.0 * B * mac / (PEAK_FP32_TFLOPS * 1e12)) if amp \
                    else F / (peak_mm * 1e12)
                t_m = A / (PEAK_HBM_TBS * 1e12)
                ts = ms_per_step * 1e-3
                out["step_roofline"] = {"flops_per_step": F, "bytes_per_step": A, "t_roof_ms": max(t_c, t_m) * 1e3,
                                        "bound": "mfma" if t_c >= t_m else "hbm",
                                        "frac_of_roof": max(t_c, t_m) / ts, "mfma_frac": t_c / ts, "hbm_frac": t_m / ts}
                # the kernel classes of a step run one after the other on one stream: the sum of every class's OWN roof (matrix pipe at
                # the multiplications it executes, or 8 TB/s) is the time a step of this structure would take with every kernel at its
                # roof - a tighter yardstick than max(F / peak, A / 8 TB/s), which lets the HBM-bound passes hide behind the convolutions
                cls_roof = sum(c["roof_ms"] for c in classes.values())
                out["step_roofline"]["sum_of_class_roofs_ms"] = cls_roof
                out["step_roofline"]["frac_of_class_roofs"] = cls_roof / (ts * 1e3)
                if mb is not None:
                    out["step_roofline"]["measured_hbm_bytes_per_step"] = mb      # rocprofv3 FETCH_SIZE + WRITE_SIZE, all kernels
                    out["step_roofline"]["measured_hbm_frac"] = mb / ts / (PEAK_HBM_TBS * 1e12)
            hist = buf.buf[:buf.n_written].cpu()
            out["final_stats"] = {k: float(hist[-1, j]) for j, k in enumerate(buf.names)}
            return out
        return None

    out = measure(args.amp)
    # The reduced-precision line (reference default `use_amp: true`, configs/base/resnet18/fixmatch.yaml:7) is measured in the
    # same process after the headline region, with the same protocol, and reported as a sub-record: the headline metric /
    # dtype / value stay those of the region the flags select (fp32 by default).
    sub = None
    if not args.amp and not args.no_amp_record:
        gc.collect(); torch.cuda.empty_cache()
        sub = measure(True)
    if rank == 0:
        if sub is not None:
            out["amp"] = {k: sub[k] for k in ("dtype", "ms_per_step", "device_ms_per_step", "host_ms_per_step", "steps", "warmup", "roofline",
                                              "kernel_classes", "step_roofline", "final_stats") if k in sub}
            out["amp"]["windows_per_s"] = sub["value"]
            out["amp"]["note"] = ("bf16 student pass (bf16 storage + v_mfma_f32_32x32x16_bf16), fp32 teacher / stem / losses; same "
                                  "workload, same process, measured after the headline region; parity: pinned block by block to the "
                                  "reference executed under PyTorch's CPU bf16 autocast (tests/test_ampfix_gpu.py, DESIGN.md section 6 N4)")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(C, L)
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
