"""Supervised plugin + the pieces every plugin shares (``src/algorithms/base.py``):
``init_model_from_cfg`` (registry-driven model factory), ``train_one_epoch``, ``evaluate``,
``train(config)``, ``test(config)``.  All arithmetic runs in the HIP library; this file is
orchestration."""
from __future__ import annotations

import datetime
import json
import os
import time
from typing import Iterable, Optional

import numpy as np
import torch
import yaml

import models.backbones as backbones
import models.decode_heads as decode_heads
import utils.lr_sched as lr_sched
import utils.misc as misc
from models.encoder_decoder import CrossEntropyLoss, EncoderDecoder
from ssecg import amp as SAMP
from ssecg import augment as SA
from ssecg import functional as SF
from ssecg import ops
from ssecg.parallel import DataParallel, unwrap
from utils.misc import NativeScalerWithGradNormCount as NativeScaler
from utils.optimizer import get_optimizer_from_config
from utils.perf_metrics import build_metric_fn, is_best_metric
from utils.semi_dataset import build_seg_dataset, device_prefetch, get_dataloader

_AMP_NOTED = [False]


def set_amp(use_amp, *models):
    """``use_amp`` (the reference's autocast switch, ``src/algorithms/fixmatch.py:97``): True selects the bf16 path for the
    train-mode student forward/backward of every given model (``ssecg.amp``: bf16 storage + bf16 MFMA, fp32 master weights,
    statistics and losses); False the fp32 path.  The teacher / pseudo-label passes of the training steps are outside autocast in
    the reference and fp32 here either way; ``evaluate()`` is INSIDE autocast in the reference (``base.py:202``) and selects its own
    16-bit eval path per call (``ssecg.amp.eval_autocast``)."""
    if use_amp and not _AMP_NOTED[0]:
        _AMP_NOTED[0] = True
        print("use_amp: true -> the student's train-mode pass runs on the bf16 path (bf16 storage + bf16 MFMA, fp32 master "
              "weights / statistics / losses; teacher / pseudo-label passes fp32, evaluate() on the 16-bit eval path).  The reference's autocast is fp16 on CUDA; this "
              "path is pinned to the reference executed under PyTorch's CPU bf16 autocast (block outputs to isolated 1-ulp "
              "flips, gradients 1e-2; the fp32 classifier tail and fp32 weight gradients are documented deviations - DESIGN.md "
              "section 6).  Set use_amp: false for the fp32 path that is pinned to the reference at 1e-4.", flush=True)
    for m in models:
        if m is not None:
            SAMP.enable(unwrap(m), bool(use_amp))


def init_model_from_cfg(config, train=True):
    """``src/algorithms/base.py:32-80``: backbone and head come from the registries by YAML key."""
    backbone_name, backbone_kwargs = list(config['backbone'].items())[0]
    assert backbone_name in backbones.__dict__, f"Unsupported model name: {backbone_name}"
    backbone = backbones.__dict__[backbone_name](**backbone_kwargs)
    decoder_name, decoder_kwargs = list(config['decode_head'].items())[0]
    assert decoder_name in decode_heads.__dict__, f"Unsupported decode head name: {decoder_name}"
    decoder = decode_heads.__dict__[decoder_name](**decoder_kwargs)
    if config.get('auxiliary_heads', None) and train:
        raise NotImplementedError("auxiliary heads are dead code in the reference (SURVEY.md Q6)")
    return EncoderDecoder(backbone=backbone, decode_head=decoder, decode_head_loss=CrossEntropyLoss(),
                          use_latent_projection=config.get('use_latent_projection', False))


def make_log_writer(output_dir):
    try:
        from torch.utils.tensorboard import SummaryWriter
        return SummaryWriter(log_dir=output_dir)
    except Exception:  # tensorboard is optional
        return None


def _log_scalars(log_writer, rows, first_step, num_steps, epoch, lrs, accum_iter):
    if log_writer is None:
        return
    for i, row in enumerate(rows):
        step = first_step + i
        if (step + 1) % accum_iter:
            continue
        x = int((epoch + step / num_steps) * 1000)  # "epoch_1000x" axis
        for k, v in row.items():
            log_writer.add_scalar(k, v, x)
        log_writer.add_scalar('lr', lrs[step], x)


def step_graph_for(holder, key, whole_step, config, accum_iter):
    """``train.hip_graph: true``: the plugin's whole step (passes, losses, backward, GradScaler update, optimiser[, EMA]) as ONE
    HIP graph after two eager steps (``ssecg/graph.py``) -> a callable with ``whole_step``'s signature, or None (eager loop).
    The graph lives on ``holder`` (the model) across epochs and is rebuilt when ``key`` (the objects the step closes over)
    changes.  Distributed runs (round 6): over RCCL (backend ``nccl``) with this library's reducer the step is captured WITH its
    collectives - ProcessGroupNCCL's all-reduces are capturable, the reducer's and SyncBatchNorm's ``work.wait()`` become stream
    dependencies of the graph, the bucket / collective order is a pure function of the model - so the reference's shipped
    operating point (16 windows per GPU on several GPUs, configs/base/resnet18/fixmatch.yaml:86, scripts/train.sh:108-141) is not
    host-bound either.  gloo (CPU-staged collectives) and torch's own DistributedDataParallel reducer keep the eager loop, as
    does gradient accumulation."""
    if config.get('hip_graph', False) is not True or accum_iter != 1 or not dist_graph_ok(holder):
        return None
    g = getattr(holder, '_ssecg_step_graph', None)
    if g is None or g.owner != key:
        from ssecg.graph import StepGraph
        if g is not None:
            g.release()          # the objects the old graph closed over are gone (new optimiser / teacher / threshold)
        g = StepGraph(whole_step)
        g.owner = key
        holder._ssecg_step_graph = g
    return g


_DIST_GRAPH_NOTED = [False]


def dist_graph_ok(holder=None, backend=None, reducer=None):
    """May a step be captured into a HIP graph in this process?  Yes without ``torch.distributed``; under it only over RCCL (``nccl``)
    and not through torch's DistributedDataParallel (its reducer has its own graph-capture protocol; ``ddp.reducer: torch``)."""
    why = None
    if backend is None and misc.is_dist_avail_and_initialized():
        backend = torch.distributed.get_backend()
    if backend is not None and backend != 'nccl':
        why = f"backend {backend}: its collectives are staged through the host and cannot be captured"
    elif reducer == 'torch' or isinstance(holder, torch.nn.parallel.DistributedDataParallel):
        why = "ddp.reducer: torch - DistributedDataParallel's own reducer is not captured"
    if why is not None and not _DIST_GRAPH_NOTED[0]:
        _DIST_GRAPH_NOTED[0] = True
        print(f"train.hip_graph: eager loop ({why})", flush=True)
    return why is None


def drop_step_graph(holder):
    """A training stage ends (ST++ stages, end of ``train()``): release the stage's HIP graph - its memory pool and the tensors
    it keeps alive - and break the holder -> graph -> step closure -> holder cycle, so the stage's model can be freed before the
    next stage captures its own graph."""
    g = getattr(holder, '_ssecg_step_graph', None)
    if g is not None:
        g.release()
        holder._ssecg_step_graph = None


def train_one_epoch(model: torch.nn.Module, data_loader: Iterable, optimizer: torch.optim.Optimizer,
                    device: torch.device, epoch: int, loss_scaler, log_writer=None, use_amp=True,
                    config: Optional[dict] = None):
    """Supervised epoch (``src/algorithms/base.py:83-181``); returns ``{'lr', 'loss'}`` global averages."""
    print_freq = 20
    accum_iter = config.get('accum_iter', 1)
    max_norm = config.get('max_norm', None)
    set_amp(use_amp, model)
    metric_logger = misc.MetricLogger(delimiter="  ")
    metric_logger.add_meter('lr', misc.SmoothedValue(window_size=1, fmt='{value:.6f}'))
    header = 'Epoch: [{}]'.format(epoch)
    model.train()
    optimizer.zero_grad()
    num_steps = len(data_loader)
    buf = misc.DeviceMetricBuffer(['loss'], num_steps, device)
    lrs, logged = [], [0]

    def whole_step(inputs, labels):
        loss = model(inputs, labels, return_loss=True)['loss']
        loss_scaler(loss, optimizer, clip_grad=max_norm, parameters=model.parameters(), update_grad=True)
        optimizer.zero_grad()
        return loss.detach().reshape(1)

    graphed = step_graph_for(model, (id(optimizer), id(loss_scaler), max_norm, bool(use_amp)), whole_step, config, accum_iter)

    def flush():
        rows = buf.flush(metric_logger)   # every rank reduces; only add_scalar is gated on the writer
        _log_scalars(log_writer, rows, logged[0], num_steps, epoch, lrs, accum_iter)
        logged[0] += len(rows)

    for data_iter_step, samples in enumerate(metric_logger.log_every(
            device_prefetch(data_loader, device, config.get('device_prefetch', True)), print_freq, header, on_print=flush)):
        if data_iter_step % accum_iter == 0:
            lr_sched.adjust_learning_rate(optimizer, data_iter_step / num_steps + epoch, config)
        inputs = samples['ecg'].to(device, non_blocking=True)
        labels = samples['target'].to(device, non_blocking=True)
        if graphed is not None:
            buf.push(graphed(inputs, labels))
        else:
            results = model(inputs, labels, return_loss=True)
            loss = results['loss']
            buf.push(loss.detach().reshape(1))
            loss_scaler(loss / accum_iter if accum_iter != 1 else loss, optimizer, clip_grad=max_norm,
                        parameters=model.parameters(), update_grad=(data_iter_step + 1) % accum_iter == 0)
            if (data_iter_step + 1) % accum_iter == 0:
                optimizer.zero_grad()
        lr = max(g["lr"] for g in optimizer.param_groups)
        lrs.append(lr)
        metric_logger.update(lr=lr)
    flush()
    metric_logger.synchronize_between_processes()
    print('Averaged stats:', metric_logger)
    return {k: meter.global_avg for k, meter in metric_logger.meters.items()}


@torch.no_grad()
def evaluate(model: torch.nn.Module, data_loader: Iterable, device: torch.device, metric_fn=None, use_amp=True,
             return_outputs: bool = True):
    """``src/algorithms/base.py:184-245``.  Eval-mode forward, argmax [+ softmax] in one kernel, per-record confusion counts on the
    device.  ``use_amp`` (round 6): the reference runs this forward INSIDE ``torch.cuda.amp.autocast(enabled=use_amp)``
    (``base.py:202``) - True = the 16-bit eval path (``ssecg.amp.eval_autocast``: bf16 convolutions / BatchNorm / residual sums
    with autocast's rounding placement on the running statistics; the 1x1 classifier, interpolation, loss and soft-max stay fp32,
    the train path's documented deviation), pinned to the reference's real ``evaluate(use_amp=True)`` under CPU bf16 autocast
    (``tests/golden/ampfix_eval_*``); False = fp32 with BN folded into the conv epilogues.  Launches are never K-split here: a
    record's logits do not depend on how many records share its batch.
    Only the (B, K, K) int32 counts cross ranks (the reference
    all-gathers the (B, K, L) probabilities and one-hot labels to every rank and feeds torchmetrics on the CPU), and the
    per-batch losses stay on the device until ONE read at the end (the reference calls ``.item()`` per batch).
    -> (valid_stats, metric_dict, outputs, labels) as the reference returns them (outputs = softmax probabilities
    (R, K, L), labels = one-hot (R, K, L), both on the host).  ``return_outputs=False`` - what every ``train()`` loop
    uses, since it discards them - skips the probabilities, their all-gather and the host copies: outputs = labels = None."""
    model.eval()
    metric_logger = misc.MetricLogger(delimiter="  ")
    outs, labs = [], []
    losses, counts_n = [], []
    for samples in metric_logger.log_every(device_prefetch(data_loader, device), 10, 'Eval:'):
        inputs = samples['ecg'].to(device, non_blocking=True)
        labels = samples['target'].to(device, non_blocking=True)
        with SAMP.eval_autocast(unwrap(model), bool(use_amp)), ops.ksplit_disabled():
            results = model(inputs, labels, return_loss=True)
        logits = results['seg_logits']
        K = logits.shape[1]
        if metric_fn is None:
            metric_fn, _ = build_metric_fn({'task': 'segmentation', 'num_classes': K, 'target_metrics': ['MeanIoU']})
        _, pred, prob = SF.pseudo_label(logits, want_prob=return_outputs)
        counts = SF.seg_confusion(pred, labels, K)
        metric_fn.update_counts(misc.concat_all_gather(counts))
        losses.append(results['loss'].detach().reshape(1))
        counts_n.append(inputs.size(0))
        if return_outputs:
            prob = misc.concat_all_gather(prob)
            labels = misc.concat_all_gather(labels)
            outs.append(prob.cpu())
            labs.append(torch.nn.functional.one_hot(labels, num_classes=K).movedim(-1, 1).cpu())
    if losses:
        host = torch.cat(losses).cpu().tolist()   # the only device read of the loop
        for v, n in zip(host, counts_n):
            metric_logger.meters['loss'].update(v, n=n)
    metric_logger.synchronize_between_processes()
    valid_stats = {k: meter.global_avg for k, meter in metric_logger.meters.items()}
    metric_dict = {}
    for k, v in metric_fn.compute().items():
        v = v.tolist()
        if isinstance(v, list):
            for i, vi in enumerate(v):
                metric_dict[f"{k}_{i}"] = vi
        else:
            metric_dict[k] = v
    print("* " + "  ".join(f"{k}: {v:.3f}" for k, v in metric_dict.items()) + f"  loss: {valid_stats['loss']:.3f}")
    metric_fn.reset()
    if not return_outputs:
        return valid_stats, metric_dict, None, None
    return valid_stats, metric_dict, torch.cat(outs, dim=0), torch.cat(labs, dim=0)


# ----------------------------------------------------------------------------- shared train()/test() scaffolding
def setup_run(config):
    misc.init_distributed_mode(config['ddp'])
    print(f'job dir: {os.path.dirname(os.path.realpath(__file__))}')
    print(yaml.dump(config, default_flow_style=False, sort_keys=False))
    device = torch.device(config['device'])
    seed = config['seed'] + misc.get_rank()
    torch.manual_seed(seed)
    np.random.seed(seed)
    SA.configure(config.get('dataset', {}), seed=seed)   # dataset.device_augment: strong view made on the GPU
    resolve_hip_graph(config)
    return device


def resolve_hip_graph(config):
    """``train.hip_graph`` absent or ``auto`` (the default since round 5): replay the whole step as one HIP graph when the run is
    one the eager loop cannot feed - at most 128 windows per loader per GPU (the reference ships ``batch_size: 16``,
    configs/base/resnet18/fixmatch.yaml:86: ~330 launches of a few microseconds each, host-bound at 5.5 ms/step eagerly
    against 3.1 ms replayed, profiles/r05_graph_bench.txt), no gradient accumulation; one GPU, or (round 6) several over RCCL
    (``ddp.dist_backend: nccl``, this library's reducer): the collectives are captured with the step (``dist_graph_ok``).
    Replayed steps are bit-identical to eager ones (tests/test_graph_gpu.py, tests/test_ddp_gpu.py); a capture that fails
    falls back to the eager loop in the same process.  ``true`` / ``false`` in the YAML are taken as given."""
    tr = config.setdefault('train', {})
    if tr.get('hip_graph', 'auto') != 'auto':
        return
    bs = int((config.get('dataloader') or {}).get('batch_size', 0) or 0)
    ddp = config['ddp']
    dist_ok = (not ddp['distributed']) or (ddp.get('dist_backend', 'nccl') == 'nccl' and ddp.get('reducer', 'ssecg') != 'torch')
    on = (str(config.get('device', 'cuda')) != 'cpu' and 0 < bs <= 128 and dist_ok and int(tr.get('accum_iter', 1) or 1) == 1)
    tr['hip_graph'] = on
    if on:
        where = "per GPU over RCCL (collectives captured with the step)" if ddp['distributed'] else "on one GPU"
        print(f"train.hip_graph: auto -> on (batch_size {bs} per loader {where}: the step is replayed as one HIP graph after two "
              "eager steps; set train.hip_graph: false for the eager loop)", flush=True)


def output_dir_and_writer(config):
    if misc.is_main_process() and config.get('output_dir'):
        output_dir = os.path.join(config['output_dir'], config['exp_name'])
        os.makedirs(output_dir, exist_ok=True)
        return output_dir, make_log_writer(output_dir)
    return None, None


def build_model(config, device):
    model = init_model_from_cfg(config)
    if config.get('mode', 'scratch') != "scratch":
        checkpoint = torch.load(config['pretrained_backbone'], map_location='cpu', weights_only=False)
        print(f"Load backbone from {config['pretrained_backbone']}")
        msg = model.backbone.load_state_dict(checkpoint['model'], strict=False)
        print(msg)
        assert set(msg.missing_keys).issubset({'mask_embedding', 'head.weight', 'head.bias'})
        if config['mode'] == "freeze_backbone":
            raise NotImplementedError("freeze_backbone is outside the hot path")
    return model.to(device)


def metrics_for(config):
    """The validation metric collection named by ``config['metric']`` (``build_metric_fn``); MeanIoU over the head's
    classes when the YAML has no metric section."""
    if config.get('metric'):
        return build_metric_fn(config['metric'])[0]
    K = list(config['decode_head'].values())[0]['num_classes']
    return build_metric_fn({'task': 'segmentation', 'num_classes': K, 'target_metrics': ['MeanIoU']})[0]


def resolve_lr(config):
    eff = config['dataloader']['batch_size'] * config['train']['accum_iter'] * misc.get_world_size()
    if config['train']['lr'] is None:
        config['train']['lr'] = config['train']['blr'] * eff / 256
    print(f"base lr: {config['train']['lr'] * 256 / eff}")
    print(f"actual lr: {config['train']['lr']}")
    print(f"accumulate grad iterations: {config['train']['accum_iter']}")
    print(f"effective batch size: {eff}")


def wrap_ddp(config, model):
    """SyncBN conversion + DDP (``src/algorithms/fixmatch.py:288-296``): gradients are all-reduced by RCCL
    in DDP's buckets, overlapped with the backward; the fused units all-reduce the BN sums themselves."""
    if not config['ddp']['distributed']:
        return model, model
    if config['ddp'].get('sync_bn', True):
        model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
    on_gpu = next(model.parameters()).is_cuda
    # 16 MB of fp32 gradients: 4 MB buckets start their ring all-reduce while the backward is still in layer3..1 (the
    # default 25 MB cap would put everything in one bucket that is reduced after the last kernel)
    # With SyncBN every rank computes the same running statistics from the all-reduced sums, so DDP's per-forward
    # broadcast of the buffers from rank 0 would only re-send identical values; without SyncBN it is kept (the reference's
    # ranks then follow rank 0's statistics, fixmatch.py:292-296 with DDP defaults).
    sync_bn = config['ddp'].get('sync_bn', True)
    # ssecg.parallel.DataParallel: DDP's contract with one multi-tensor staging launch per bucket instead of one per parameter, and
    # ``param.grad`` living inside the bucket (MEASURED on one rank with the collectives forced, tools/dist_overhead.sh: DESIGN.md
    # section 4).  ``ddp.reducer: torch`` keeps torch's DistributedDataParallel (gradient_as_bucket_view) for comparison.
    if config['ddp'].get('reducer', 'ssecg') == 'torch':
        ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[config['ddp']['gpu']] if on_gpu else None,
                                                        bucket_cap_mb=config['ddp'].get('bucket_cap_mb', 4),
                                                        broadcast_buffers=not sync_bn,
                                                        gradient_as_bucket_view=config['ddp'].get('gradient_as_bucket_view', True))
    else:
        ddp = DataParallel(model, bucket_cap_mb=config['ddp'].get('bucket_cap_mb', 4), broadcast_buffers=not sync_bn)
    return ddp, ddp.module


def epoch_tail(config, output_dir, log_writer, epoch, model_without_ddp, optimizer, loss_scaler, train_stats,
               valid_stats, metrics, best, model_ema=None, metric_fn=None):
    """Best-loss / best-metric checkpoints, TensorBoard, log.txt (``src/algorithms/fixmatch.py:345-401``)."""
    curr_loss = valid_stats['loss']
    if output_dir and curr_loss < best['loss']:
        best['loss'] = curr_loss
        misc.save_model(config, os.path.join(output_dir, 'best-loss.pth'), epoch, model_without_ddp, optimizer,
                        loss_scaler, metrics={'loss': curr_loss, **metrics}, model_ema=model_ema)
    for name, metric_class in (metric_fn or {}).items():
        value = metrics[name]
        print(f"{name}: {value:.3f}")
        best.setdefault(name, -float('inf') if metric_class.higher_is_better else float('inf'))
        if output_dir and is_best_metric(metric_class, best[name], value):
            best[name] = value
            misc.save_model(config, os.path.join(output_dir, f'best-{name}.pth'), epoch, model_without_ddp, optimizer,
                            loss_scaler, metrics={'loss': curr_loss, **metrics}, model_ema=model_ema)
        print(f"Best {name}: {best.get(name, value):.3f}")
    if log_writer is not None:
        log_writer.add_scalar('perf/valid_loss', curr_loss, epoch)
        for name, value in metrics.items():
            log_writer.add_scalar(f'perf/{name}', value, epoch)
    log_stats = {**{f'train_{k}': v for k, v in train_stats.items()}, **{f'valid_{k}': v for k, v in valid_stats.items()},
                 **metrics, 'epoch': epoch}
    if output_dir and misc.is_main_process():
        if log_writer is not None:
            log_writer.flush()
        with open(os.path.join(output_dir, 'log.txt'), mode='a', encoding="utf-8") as f:
            f.write(json.dumps(log_stats) + '\n')


def train(config):
    device = setup_run(config)
    dataset_train = build_seg_dataset(config['dataset'], split='train_labeled')
    dataset_valid = build_seg_dataset(config['dataset'], split='valid')
    loader_train = get_dataloader(dataset_train, is_distributed=config['ddp']['distributed'], mode='train', **config['dataloader'])
    loader_valid = get_dataloader(dataset_valid, is_distributed=config['ddp']['distributed'], mode='valid', **config['dataloader'])
    output_dir, log_writer = output_dir_and_writer(config)
    model = build_model(config, device)
    print(f"Model = {model}")
    resolve_lr(config)
    model, model_without_ddp = wrap_ddp(config, model)
    optimizer = get_optimizer_from_config(config['train'], model_without_ddp.parameters())
    print(f"Optimizer = {optimizer}")
    loss_scaler = NativeScaler()
    misc.load_model(config, model_without_ddp, optimizer, loss_scaler)
    best = {'loss': float('inf')}
    metric_fn = metrics_for(config)
    num_epochs = config['train']['epochs']
    use_amp = config.get('use_amp', True)
    print(f"Start training for {num_epochs} epochs")
    start_time = time.time()
    for epoch in range(config['start_epoch'], num_epochs):
        if config['ddp']['distributed']:
            loader_train.sampler.set_epoch(epoch)
        train_stats = train_one_epoch(model, loader_train, optimizer, device, epoch, loss_scaler, log_writer,
                                      use_amp=use_amp, config=config['train'])
        valid_stats, metrics, _, _ = evaluate(model, loader_valid, device, metric_fn, use_amp=use_amp, return_outputs=False)
        epoch_tail(config, output_dir, log_writer, epoch, model_without_ddp, optimizer, loss_scaler, train_stats,
                   valid_stats, metrics, best, metric_fn=metric_fn)
    print(f'Training time {datetime.timedelta(seconds=int(time.time() - start_time))}')
    if log_writer is not None:
        log_writer.close()


def test(config):
    """``src/algorithms/base.py:442-499``: reload ``test.model_path`` or ``best-<target_metric>.pth``, evaluate the test
    split and write the reference's three artefacts: ``test_metrics.csv`` (one row, ``%.4f``), ``test_outputs.npy``
    (softmax probabilities (R, K, L)) and ``test_labels.npy`` (one-hot labels (R, K, L))."""
    output_dir = os.path.join(config['output_dir'], config['exp_name'])
    os.makedirs(output_dir, exist_ok=True)
    device = torch.device(config['device'])
    dataset_test = build_seg_dataset(config['dataset'], split='test')
    loader = get_dataloader(dataset_test, is_distributed=False, mode='test', **config['dataloader'])
    model = init_model_from_cfg(config, train=False)
    tcfg = config.get('test') if isinstance(config.get('test'), dict) else {}
    if tcfg.get('model_path'):
        ckpt_path = tcfg['model_path']
    else:
        ckpt_path = os.path.join(output_dir, f"best-{tcfg.get('target_metric', 'loss')}.pth")
    assert os.path.exists(ckpt_path), f"Checkpoint not found: {ckpt_path}"
    ckpt = torch.load(ckpt_path, map_location='cpu', weights_only=False)
    state = {k: v for k, v in ckpt['model'].items() if not k.startswith('auxiliary_head')}   # drop the auxiliary head
    print(model.load_state_dict(state))
    model.to(device)
    stats, metrics, outputs, labels = evaluate(model, loader, device, metrics_for(config), use_amp=config.get('use_amp', True),
                                               return_outputs=True)
    metrics = dict(metrics)
    metrics['loss'] = stats['loss']
    import pandas as pd
    pd.DataFrame([metrics]).to_csv(os.path.join(output_dir, 'test_metrics.csv'), index=False, float_format='%.4f')
    np.save(os.path.join(output_dir, 'test_outputs.npy'), outputs.numpy())
    np.save(os.path.join(output_dir, 'test_labels.npy'), labels.numpy())
    print('Done!')
    return metrics
