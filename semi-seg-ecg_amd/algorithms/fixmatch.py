"""FixMatch plugin (``src/algorithms/fixmatch.py``): the weak view is pseudo-labelled by the model in
eval mode (BN folded into the conv epilogues -> one kernel per conv), the student pass runs labelled +
strong views as one 2B batch, and both cross-entropy terms with the confidence mask are one fused node."""
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


def fixmatch_step(model, ecg_x, mask_x, ecg_u_w, ecg_u_s, conf_thresh):
    """Forward part of one iteration (``fixmatch.py:86-118``) -> (loss, stats[loss_total, loss_x, loss_u_s, mask_ratio])."""
    from ssecg import ops
    # one operand-refresh scope for both passes: nothing rewrites the weights between the pseudo-label pass and the student pass
    # (the optimiser's own launch ends the scope's validity: ops.weights_changed), so the Winograd / bf16 weight operands are
    # formed once per step instead of once per forward
    with ops.model_scope(), ops.PassOverlap(ecg_x.size(0), ecg_x.device, model) as ov:   # the pseudo-label pass on a side stream
        with ov.teacher(), torch.no_grad():
            model.eval()
            pred_u_w = model(ecg_u_w, return_loss=False)['seg_logits']
            conf_u_w, mask_u_w, _ = SF.pseudo_label(pred_u_w)
        model.train()
        logits = model(ops.batch_pair(ecg_x, ecg_u_s), return_loss=False)['seg_logits']
    return SF.fixmatch_loss(logits, ecg_x.size(0), mask_x, mask_u_w, conf_u_w, conf_thresh)


def train_one_epoch(model: torch.nn.Module, labeled_data_loader: Iterable, unlabeled_data_loader: Iterable,
                    optimizer: torch.optim.Optimizer, device: torch.device, epoch: int, loss_scaler, log_writer=None,
                    use_amp=True, config: Optional[dict] = None):
    """FixMatch epoch; returns global averages of ``lr, loss_total, loss_x, loss_u_s, mask_ratio``."""
    print_freq = 20
    accum_iter = config.get('accum_iter', 1)
    max_norm = config.get('max_norm', None)
    set_amp(use_amp, model)
    metric_logger = misc.MetricLogger(delimiter="  ")
    metric_logger.add_meter('lr', misc.SmoothedValue(window_size=1, fmt='{value:.6f}'))
    header = 'Epoch: [{}]'.format(epoch)
    model.train()
    optimizer.zero_grad()
    num_steps = len(unlabeled_data_loader)
    assert len(labeled_data_loader) == num_steps, "The number of labeled and unlabeled data should be the same"
    buf = misc.DeviceMetricBuffer(['loss_total', 'loss_x', 'loss_u_s', 'mask_ratio'], num_steps, device)
    lrs, logged = [], [0]

    def whole_step(ecg_x, mask_x, ecg_u_w, ecg_u_s):
        loss, stats = fixmatch_step(model, ecg_x, mask_x, ecg_u_w, ecg_u_s, config['conf_thresh'])
        loss_scaler(loss, optimizer, clip_grad=max_norm, parameters=model.parameters(), update_grad=True)
        optimizer.zero_grad()
        return stats

    # train.hip_graph: the whole step as one HIP graph after two eager steps (algorithms/base.py:step_graph_for)
    graphed = step_graph_for(model, (id(optimizer), id(loss_scaler), config['conf_thresh'], max_norm, bool(use_amp)),
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
            lr_sched.adjust_learning_rate(optimizer, data_iter_step / num_steps + epoch, config)
        ecg_x = labeled['ecg'].to(device, non_blocking=True)
        mask_x = labeled['target'].to(device, non_blocking=True)
        ecg_u_w, ecg_u_s = SA.unlabeled_views(unlabeled, device)   # host-made views, or made here from 'ecg_raw'
        if graphed is not None:
            buf.push(graphed(ecg_x, mask_x, ecg_u_w, ecg_u_s))
        else:
            loss, stats = fixmatch_step(model, ecg_x, mask_x, ecg_u_w, ecg_u_s, config['conf_thresh'])
            buf.push(stats)
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


def train(config):
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
        if dist_on:
            loader_l.sampler.set_epoch(epoch)
            loader_u.sampler.set_epoch(epoch)
        train_stats = train_one_epoch(model, loader_l, loader_u, optimizer, device, epoch, loss_scaler, log_writer,
                                      use_amp=use_amp, config=config['train'])
        valid_stats, metrics, _, _ = evaluate(model, loader_v, device, metric_fn, use_amp=use_amp, return_outputs=False)
        epoch_tail(config, output_dir, log_writer, epoch, model_without_ddp, optimizer, loss_scaler, train_stats,
                   valid_stats, metrics, best, metric_fn=metric_fn)
    print(f'Training time {datetime.timedelta(seconds=int(time.time() - start_time))}')
    if log_writer is not None:
        log_writer.close()
