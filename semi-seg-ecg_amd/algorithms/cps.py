"""Cross Pseudo Supervision plugin (``src/algorithms/cps.py``): two networks label the weak view for each other.
Per iteration: both models pseudo-label ``ecg_u_w`` in eval mode (BN folded, argmax in the softmax kernel), then each
model takes one training pass over ``cat(ecg_x, ecg_u_w)`` with the OTHER model's hard labels on the unlabelled half
(no confidence mask) and its own AdamW step.  No new arithmetic: same conv/BN/loss kernels as FixMatch."""
from __future__ import annotations

import datetime
import time
from typing import Iterable, Optional

import torch

import utils.lr_sched as lr_sched
import utils.misc as misc
from algorithms.base import (_log_scalars, build_model, epoch_tail, evaluate, init_model_from_cfg, metrics_for, set_amp,  # noqa: F401
                             output_dir_and_writer, resolve_lr, setup_run, step_graph_for, test, wrap_ddp)
from ssecg import augment as SA
from ssecg import functional as SF
from ssecg import ops
from utils.misc import NativeScalerWithGradNormCount as NativeScaler
from utils.optimizer import get_optimizer_from_config
from utils.semi_dataset import build_seg_dataset, device_prefetch, get_dataloader


def cps_pseudo_labels(model_1, model_2, ecg_u_w):
    """``cps.py:96-103``: argmax of each model's eval-mode logits on the weak view."""
    with torch.no_grad():
        model_1.eval()
        model_2.eval()
        _, mask_1, _ = SF.pseudo_label(model_1(ecg_u_w, return_loss=False)['seg_logits'])
        _, mask_2, _ = SF.pseudo_label(model_2(ecg_u_w, return_loss=False)['seg_logits'])
    return mask_1, mask_2


def cps_loss(model, ecg_x, mask_x, ecg_u_w, mask_u_w):
    """``cps.py:113-137`` -> (loss, stats[loss_total, loss_x, loss_u_s, 1]); loss = (CE_x + CE_u) / 2."""
    model.train()
    logits = model(ops.batch_pair(ecg_x, ecg_u_w), return_loss=False)['seg_logits']
    return SF.fixmatch_loss(logits, ecg_x.size(0), mask_x, mask_u_w, None, 0.0)


def train_one_epoch(model_1: torch.nn.Module, model_2: torch.nn.Module, labeled_data_loader: Iterable,
                    unlabeled_data_loader: Iterable, optimizer_1: torch.optim.Optimizer,
                    optimizer_2: torch.optim.Optimizer, device: torch.device, epoch: int, loss_scaler, log_writer=None,
                    use_amp=True, config: Optional[dict] = None):
    """CPS epoch; returns global averages of ``lr, loss_total, loss_x, loss_u_s`` (each the mean over the two models,
    ``cps.py:164-170``)."""
    print_freq = 20
    accum_iter = config.get('accum_iter', 1)
    max_norm = config.get('max_norm', None)
    set_amp(use_amp, model_1, model_2)
    metric_logger = misc.MetricLogger(delimiter="  ")
    metric_logger.add_meter('lr', misc.SmoothedValue(window_size=1, fmt='{value:.6f}'))
    header = 'Epoch: [{}]'.format(epoch)
    model_1.train()
    model_2.train()
    optimizer_1.zero_grad()
    optimizer_2.zero_grad()
    num_steps = len(unlabeled_data_loader)
    assert len(labeled_data_loader) == num_steps, "The number of labeled and unlabeled data should be the same"
    buf = misc.DeviceMetricBuffer(['loss_total', 'loss_x', 'loss_u_s'], num_steps, device)
    lrs, logged = [], [0]

    def whole_step(ecg_x, mask_x, ecg_u_w):
        mask_u_w_1, mask_u_w_2 = cps_pseudo_labels(model_1, model_2, ecg_u_w)
        step_stats = None
        for model, optimizer, mask_u_w in ((model_1, optimizer_1, mask_u_w_2), (model_2, optimizer_2, mask_u_w_1)):
            loss, stats = cps_loss(model, ecg_x, mask_x, ecg_u_w, mask_u_w)
            loss_scaler(loss, optimizer, clip_grad=max_norm, parameters=model.parameters(), update_grad=True)
            optimizer.zero_grad()
            step_stats = stats[:3] if step_stats is None else step_stats + stats[:3]
        return step_stats * 0.5

    # train.hip_graph: both models' passes, backwards and optimiser steps as one HIP graph (algorithms/base.py:step_graph_for)
    graphed = step_graph_for(model_1, (id(model_2), id(optimizer_1), id(optimizer_2), id(loss_scaler), max_norm, bool(use_amp)),
                             whole_step, config, accum_iter)

    def flush():
        rows = buf.flush(metric_logger)   # every rank reduces; only add_scalar is gated on the writer
        _log_scalars(log_writer, rows, logged[0], num_steps, epoch, lrs, accum_iter)
        logged[0] += len(rows)

    for data_iter_step, (labeled, unlabeled) in enumerate(metric_logger.log_every(
            zip(device_prefetch(labeled_data_loader, device, config.get('device_prefetch', True)),
                device_prefetch(unlabeled_data_loader, device, config.get('device_prefetch', True))),
            print_freq, header, length=num_steps, on_print=flush)):
        if data_iter_step % accum_iter == 0:
            lr_sched.adjust_learning_rate(optimizer_1, data_iter_step / num_steps + epoch, config)
            lr_sched.adjust_learning_rate(optimizer_2, data_iter_step / num_steps + epoch, config)
        ecg_x = labeled['ecg'].to(device, non_blocking=True)
        mask_x = labeled['target'].to(device, non_blocking=True)
        ecg_u_w, _ = SA.unlabeled_views(unlabeled, device, want_strong=False)
        if graphed is not None:
            buf.push(graphed(ecg_x, mask_x, ecg_u_w))
            lr = max(g["lr"] for g in optimizer_2.param_groups)
            lrs.append(lr)
            metric_logger.update(lr=lr)
            continue
        mask_u_w_1, mask_u_w_2 = cps_pseudo_labels(model_1, model_2, ecg_u_w)
        step_stats = None
        for model, optimizer, mask_u_w in ((model_1, optimizer_1, mask_u_w_2), (model_2, optimizer_2, mask_u_w_1)):
            loss, stats = cps_loss(model, ecg_x, mask_x, ecg_u_w, mask_u_w)
            loss_scaler(loss / accum_iter if accum_iter != 1 else loss, optimizer, clip_grad=max_norm,
                        parameters=model.parameters(), update_grad=(data_iter_step + 1) % accum_iter == 0)
            if (data_iter_step + 1) % accum_iter == 0:
                optimizer.zero_grad()
            step_stats = stats[:3] if step_stats is None else step_stats + stats[:3]
        buf.push(step_stats * 0.5)
        lr = max(g["lr"] for g in optimizer_2.param_groups)
        lrs.append(lr)
        metric_logger.update(lr=lr)
    flush()
    metric_logger.synchronize_between_processes()
    print('Averaged stats:', metric_logger)
    return {k: meter.global_avg for k, meter in metric_logger.meters.items()}


def train(config):
    """``cps.py:210-416``: model_1 is the one validated and checkpointed."""
    device = setup_run(config)
    ds_u = build_seg_dataset(config['dataset'], split='train_unlabeled')
    ds_l = build_seg_dataset(config['dataset'], split='train_labeled', num_unlabeled=len(ds_u))
    ds_v = build_seg_dataset(config['dataset'], split='valid')
    dist_on = config['ddp']['distributed']
    loader_l = get_dataloader(ds_l, is_distributed=dist_on, mode='train', **config['dataloader'])
    print(f"Labeled: {len(ds_l)} samples / {len(loader_l)} batches")
    loader_u = get_dataloader(ds_u, is_distributed=dist_on, mode='train', **config['dataloader'])
    print(f"Unlabeled: {len(ds_u)} samples / {len(loader_u)} batches")
    loader_v = get_dataloader(ds_v, is_distributed=dist_on, mode='valid', **config['dataloader'])
    output_dir, log_writer = output_dir_and_writer(config)
    model_1 = build_model(config, device)
    model_2 = build_model(config, device)
    print(f"Model = {model_1}")
    resolve_lr(config)
    model_1, model_without_ddp_1 = wrap_ddp(config, model_1)
    model_2, model_without_ddp_2 = wrap_ddp(config, model_2)
    optimizer_1 = get_optimizer_from_config(config['train'], model_without_ddp_1.parameters())
    optimizer_2 = get_optimizer_from_config(config['train'], model_without_ddp_2.parameters())
    print(f"Optimizer = {optimizer_1}")
    loss_scaler = NativeScaler()
    best = {'loss': float('inf')}
    metric_fn = metrics_for(config)
    num_epochs = config['train']['epochs']
    use_amp = config.get('use_amp', True)
    print(f"Start training for {num_epochs} epochs")
    start_time = time.time()
    for epoch in range(config['start_epoch'], num_epochs):
        if dist_on:
            loader_l.sampler.set_epoch(epoch)
            loader_u.sampler.set_epoch(epoch)
        train_stats = train_one_epoch(model_1, model_2, loader_l, loader_u, optimizer_1, optimizer_2, device, epoch,
                                      loss_scaler, log_writer, use_amp=use_amp, config=config['train'])
        valid_stats, metrics, _, _ = evaluate(model_1, loader_v, device, metric_fn, use_amp=use_amp, return_outputs=False)
        epoch_tail(config, output_dir, log_writer, epoch, model_without_ddp_1, optimizer_1, loss_scaler, train_stats,
                   valid_stats, metrics, best, metric_fn=metric_fn)
    print(f'Training time {datetime.timedelta(seconds=int(time.time() - start_time))}')
    if log_writer is not None:
        log_writer.close()
