#!/usr/bin/env python3
"""Headline benchmark: ECG windows/s of the FixMatch training step on MI355X.

Workload (BASELINE.json metric / SURVEY.md §8d headline config): FixMatch, ResNet18-1D + FCNHead,
B = 512 labelled + 512 unlabelled windows per GPU per step (weak + strong views), 12 leads, L = 2000,
fp32, AdamW, SyncBN + DDP when N > 1.  A "step" = teacher pass (eval) + student pass over 2B windows +
both losses + backward + AdamW; inputs are synthetic and resident in HBM before the timed region.

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

Prints ONE JSON line on rank 0 (contract in the task statement) including
  "roofline"     : the dominant kernel's achieved TFLOP/s, timed with HIP events on the launch stream
                   during one extra instrumented step (outside the timed region),
  "cpu_baseline" : the oracle (oracle/torch_ref.py, a CPU restatement) timed on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(ROOT, "semi-seg-ecg_amd")
for p in (ROOT, SRC):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_FP32_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 matrix == vector peak
PEAK_HBM_TBS = 8.0
MAC_BASE, MAC_PER_LEAD = 353_828_352, 448_000   # conv MACs per window, forward, L=2000
BYTES_PER_B = 62.2e6                             # algorithmic HBM bytes per unit of B per FixMatch step
PARAM_BYTES = 0.2e9


def model_config(C):
    return {"backbone": {"resnet18": dict(num_leads=C, num_stages=4, out_indices=[0, 1, 2, 3], dilations=[1, 1, 1, 1],
                                           strides=[1, 2, 2, 2], deep_stem=False, avg_down=False, contract_dilation=False)},
            "decode_head": {"FCNHead": dict(in_channels=512, in_index=3, channels=128, num_convs=1, concat_input=False,
                                            dropout_ratio=0.1, num_classes=4, align_corners=False)}}


TRAIN_CFG = dict(epochs=100, accum_iter=1, warmup_epochs=10, min_lr=1e-4, lr=1e-3, weight_decay=0.05, max_norm=None,
                 optimizer="adamw", optimizer_kwargs={"betas": [0.9, 0.999]}, conf_thresh=0.80)


def synthetic_batch(B, C, L, seed, device):
    """ecg ~ N(0,1), strong = weak + 0.5 N(0,1), piecewise-constant labels; generated on the device."""
    g = torch.Generator(device=device).manual_seed(seed)
    ecg_x = torch.randn((B, C, L), generator=g, device=device)
    ecg_u_w = torch.randn((B, C, L), generator=g, device=device)
    ecg_u_s = ecg_u_w + 0.5 * torch.randn((B, C, L), generator=g, device=device)
    from ssecg import synth
    mask_x = torch.from_numpy(synth.labels(seed, 4, B, L)).to(device)
    return ecg_x, mask_x, ecg_u_w, ecg_u_s


def cpu_baseline(C, L, Bc=16, steps=4):
    """The oracle's FixMatch step on the host cores (rank 0, N = 1 only): a bounded sample of the same workload."""
    from oracle import torch_ref as O
    from ssecg import synth
    sd = O.state_from_numpy(synth.model_state(0, C))
    cfg = dict(TRAIN_CFG); cfg["betas"] = (0.9, 0.999)
    batch = {g: {k: torch.from_numpy(v) for k, v in d.items()} for g, d in synth.fixmatch_batch(1234, Bc, C, L).items()}
    opt = {}
    O.fixmatch_step(sd, opt, batch, cfg, 3.0)  # warm-up
    t0 = time.time()
    for _ in range(steps):
        O.fixmatch_step(sd, opt, batch, cfg, 3.0)
    dt = (time.time() - t0) / steps
    return {"value": Bc / dt, "unit": "windows/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle/torch_ref.fixmatch_step, B={Bc}, C={C}, L={L}, fp32, {steps} timed steps after 1 warm-up, "
                      f"{dt * 1e3:.0f} ms/step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=512, help="labelled (= unlabelled) windows per GPU per step")
    ap.add_argument("--leads", type=int, default=12)
    ap.add_argument("--length", type=int, default=2000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL over xGMI on ROCm); gloo only for rehearsals")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    if os.environ.get("SSECG_BENCH_SHARE_GPU") == "1":
        local_rank = 0  # rehearsal only: several ranks on one card (use --backend gloo; RCCL needs one GPU per rank)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    distributed = world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=args.backend, init_method="env://", world_size=world, rank=rank)

    import utils.lr_sched as lr_sched
    from algorithms.base import init_model_from_cfg
    from algorithms.fixmatch import fixmatch_step
    from ssecg import ops
    from utils.misc import DeviceMetricBuffer, NativeScalerWithGradNormCount
    from utils.optimizer import get_optimizer_from_config

    B, C, L = args.batch, args.leads, args.length
    torch.manual_seed(0)  # identical random-init weights on every rank (reference init law)
    model = init_model_from_cfg(model_config(C)).to(device)
    # SyncBN conversion + DDP exactly as the plugins' train(config) does it (algorithms/base.py:wrap_ddp)
    from algorithms.base import wrap_ddp
    model, model_without_ddp = wrap_ddp({"ddp": {"distributed": distributed, "sync_bn": True, "gpu": local_rank}}, model)
    cfg = dict(TRAIN_CFG)
    optimizer = get_optimizer_from_config(cfg, model_without_ddp.parameters())
    scaler = NativeScalerWithGradNormCount()
    ecg_x, mask_x, ecg_u_w, ecg_u_s = synthetic_batch(B, C, L, 1234 + rank, device)
    total = args.warmup + args.steps + 1
    buf = DeviceMetricBuffer(['loss_total', 'loss_x', 'loss_u_s', 'mask_ratio'], total, device)

    def one_step(i):
        lr_sched.adjust_learning_rate(optimizer, 10.0 + i / 1000.0, cfg)
        loss, stats = fixmatch_step(model, ecg_x, mask_x, ecg_u_w, ecg_u_s, cfg['conf_thresh'])
        buf.push(stats)
        scaler(loss, optimizer, clip_grad=None, parameters=model.parameters(), update_grad=True)
        optimizer.zero_grad()

    for i in range(args.warmup):
        one_step(i)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(args.steps):
        one_step(args.warmup + i)
    ev1.record()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    t = torch.tensor([wall], dtype=torch.float64, device=device)
    if distributed:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall = t.item()

    # ---- one extra instrumented step: HIP events around every conv launch on the launch stream ----
    ops.PROFILE = []
    one_step(total - 1)
    torch.cuda.synchronize()
    prof, ops.PROFILE = ops.PROFILE, None
    per = {}
    for name, flops, e0, e1, nbytes in prof:
        d = per.setdefault(name, [0.0, 0.0, 0, 0.0])
        d[0] += flops; d[1] += e0.elapsed_time(e1) * 1e-3; d[2] += 1; d[3] += nbytes
    dom = max(per.items(), key=lambda kv: kv[1][1])
    conv_time = sum(v[1] for v in per.values())

    if rank == 0:
        ms_per_step = wall / args.steps * 1e3
        value = world * B * args.steps / wall
        mac = MAC_BASE + MAC_PER_LEAD * C if L == 2000 else None
        out = {
            "metric": f"ECG windows/sec (FixMatch step, ResNet18-1D + FCNHead, B={B}/GPU, L={L}, {C}-lead, fp32; whole job = "
                      f"per-GPU x n_gpus)",
            "value": value, "unit": "windows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"FixMatch step, {B} labelled + {B} unlabelled windows/GPU (weak+strong views), "
                                   f"{C} leads, L={L}, ResNet18-1D + FCNHead, AdamW, random-init weights",
                       "global_batch": world * B, "parallelism": f"dp{world}" + ("+syncbn" if distributed else "")},
            "per_gpu_windows_per_s": value / world,
            "device_ms_per_step": dev_ms / args.steps,
        }
        dname, (dfl, dsec, dn, dby) = dom
        ach = dfl / dsec / 1e12
        out["roofline"] = {"bound": "mfma", "kernel": dname, "achieved": ach, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                           "frac": ach / PEAK_FP32_TFLOPS, "traffic": None, "launches_per_step": dn,
                           "avg_launch_ms": dsec / dn * 1e3,
                           "algorithmic_flops_per_launch": dfl / dn, "algorithmic_bytes_per_launch": (dby / dn) if dby else None,
                           "all_conv_kernels": {k: {"tflops": v[0] / v[1] / 1e12, "ms_per_step": v[1] * 1e3, "launches": v[2]}
                                                for k, v in sorted(per.items())},
                           "conv_ms_per_step": conv_time * 1e3}
        # measured HBM traffic of the dominant kernel: PMC passes cannot run inside this process, so the per-launch figure
        # comes from the committed rocprofv3 summary of this same command (profiles/README.md), if it has this kernel
        try:
            import glob
            tfile = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_traffic.json")))[-1]
            tk = json.load(open(tfile))["kernels"].get(dname.split(" ")[0] if " (" in dname else dname)
            if tk:
                out["roofline"]["traffic"] = tk["read_bytes"] + tk["write_bytes"]
                out["roofline"]["traffic_detail"] = {"read_bytes": tk["read_bytes"], "write_bytes": tk["write_bytes"],
                                                     "source": os.path.relpath(tfile, os.path.dirname(os.path.abspath(__file__)))}
        except Exception:
            pass
        try:   # step-level measured HBM traffic from the same committed PMC summary (all kernels x launches per step)
            allk = json.load(open(tfile))["kernels"]
            hb = sum((v["read_bytes"] + v["write_bytes"]) * v["launches_per_step"] for v in allk.values())
            out["roofline"]["measured_hbm_bytes_per_step"] = hb
        except Exception:
            pass
        if "wino" in dname:
            # algorithmic FLOPs of the direct 3-tap convolution / time; the Winograd F(2,3) kernel executes
            # 2/3 of those multiplications on the matrix pipe, so the pipe's own rate is 2/3 of `achieved`
            out["roofline"]["executed_tflops"] = ach * 2.0 / 3.0
            out["roofline"]["note"] = ("achieved = algorithmic direct-conv FLOPs / kernel time; the kernel is Winograd F(2,3) "
                                       "(4 instead of 6 multiplications per output pair and channel pair), MFMA-executed rate "
                                       "= 2/3 of achieved")
        if mac is not None:
            # F = 14*B*MAC FLOPs (2 FLOP/MAC x [teacher B + student 2B] forward + 4 FLOP/MAC x 2B backward)
            F = 14.0 * B * mac
            A = BYTES_PER_B * B + PARAM_BYTES
            t_c, t_m = F / (PEAK_FP32_TFLOPS * 1e12), A / (PEAK_HBM_TBS * 1e12)
            ts = ms_per_step * 1e-3
            out["step_roofline"] = {"flops_per_step": F, "bytes_per_step": A, "t_roof_ms": max(t_c, t_m) * 1e3,
                                    "frac_of_roof": max(t_c, t_m) / ts, "mfma_frac": t_c / ts, "hbm_frac": t_m / ts}
            if "measured_hbm_bytes_per_step" in out["roofline"]:
                mb = out["roofline"].pop("measured_hbm_bytes_per_step")
                out["step_roofline"]["measured_hbm_bytes_per_step"] = mb      # rocprofv3 FETCH_SIZE + WRITE_SIZE, all kernels
                out["step_roofline"]["measured_hbm_frac"] = mb / ts / (PEAK_HBM_TBS * 1e12)
        hist = buf.buf[:buf.n_written].cpu()
        out["final_stats"] = {k: float(hist[-1, j]) for j, k in enumerate(buf.names)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(C, L)
        print(json.dumps(out))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
