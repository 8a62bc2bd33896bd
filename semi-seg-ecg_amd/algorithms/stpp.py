"""ST++ plugin (``src/algorithms/stpp.py``): (1) supervised training that keeps three checkpoints, (2) unlabelled
records ranked by how stably those checkpoints label them, re-training on labelled + the reliable half with a frozen
teacher's hard labels, (3) the same on the whole unlabelled set with the stage-2 model as teacher.

The per-iteration arithmetic is the FixMatch kernel set without the confidence mask (student on ``cat(ecg_x,
ecg_u_w)``, teacher in eval mode with BN folded).  The reliability score is computed from per-record confusion counts
on the device (``ssecg_seg_confusion``) instead of moving (1, K, L) one-hots of every record to the host."""
from __future__ import annotations

import datetime
import os
import time
from typing import Iterable, Optional

import numpy as np
import torch
from torch.utils.data import DataLoader

import utils.lr_sched as lr_sched
import utils.misc as misc
from algorithms.base import (_log_scalars, build_model, epoch_tail, evaluate, init_model_from_cfg, metrics_for, set_amp,  # noqa: F401
                             output_dir_and_writer, resolve_lr, setup_run, step_graph_for, test, wrap_ddp)
from algorithms.base import drop_step_graph
from algorithms.base import train_one_epoch as train_one_epoch_labeled
from ssecg import augment as SA
from ssecg import functional as SF
from ssecg import ops
from utils.misc import NativeScalerWithGradNormCount as NativeScaler
from utils.optimizer import get_optimizer_from_config
from utils.semi_dataset import build_seg_dataset, device_prefetch, get_dataloader


def calculate_miou(onehot_preds, onehot_labels, ignore_background=False):
    """``stpp.py:32-42`` (host arrays, kept for callers of the reference's helper): IoU per class pooled over every
    record passed in, 0 for an empty union, mean over classes."""
    onehot_preds, onehot_labels = np.asarray(onehot_preds), np.asarray(onehot_labels)
    if ignore_background:
        onehot_preds, onehot_labels = onehot_preds[:, 1:], onehot_labels[:, 1:]
    ious = []
    for c in range(onehot_preds.shape[1]):
        intersection = (onehot_preds[:, c] * onehot_labels[:, c]).sum()
        union = onehot_preds[:, c].sum() + onehot_labels[:, c].sum() - intersection
        ious.append(intersection / union if union > 0 else 0.0)
    return np.mean(ious)


@torch.no_grad()
def record_reliability(models, ecg):
    """(B,) float64 on the device: mean over the earlier checkpoints of mIoU(pred_k, pred_last) per record
    (``stpp.py:58-80``)."""
    preds = []
    K = None
    for model in models:
        logits = model(ecg, return_loss=False)['seg_logits']
        K = logits.shape[1]
        preds.append(SF.pseudo_label(logits)[1])
    mious = [SF.iou_from_confusion(SF.seg_confusion(p, preds[-1], K)).mean(dim=1) for p in preds[:-1]]
    return torch.stack(mious).sum(dim=0) / len(mious)


@torch.no_grad()
def select_reliable(models, dataloader, device, reference_ids=False):
    """``stpp.py:45-88`` -> (reliable_indices, unreliable_indices): records sorted by decreasing reliability (stable),
    first half reliable.  Any batch size is accepted (the reference asserts 1).

    ``reference_ids=True`` returns what the reference literally returns: its inner ``for i in range(len(onehot_preds)
    - 1)`` loop (stpp.py:72) overwrites the record index before ``id_to_reliability.append((i, reliability))``
    (stpp.py:81), so every id equals ``len(models) - 2``.  The default keeps the record indices (the evident intent)."""
    for model in models:
        model.eval()
    rel = []
    for data in dataloader:
        ecg = data['ecg'].to(device, non_blocking=True)
        rel.append(record_reliability(models, ecg))
    rel = torch.cat(rel).cpu().tolist()
    ids = [len(models) - 2] * len(rel) if reference_ids else list(range(len(rel)))
    id_to_reliability = sorted(zip(ids, rel), key=lambda elem: elem[1], reverse=True)
    half = len(id_to_reliability) // 2
    return [e[0] for e in id_to_reliability[:half]], [e[0] for e in id_to_reliability[half:]]


def stpp_step(model_student, model_teacher, ecg_x, mask_x, ecg_u_w):
    """``stpp.py:150-183`` -> (loss, stats[loss_total, loss_x, loss_u_s, 1])."""
    from ssecg import ops
    # frozen teacher and student: one operand refresh per step (see fixmatch_step); the teacher pass on a side stream
    with ops.model_scope(), ops.PassOverlap(ecg_x.size(0), ecg_x.device, model_teacher, model_student) as ov:
        with ov.teacher(), torch.no_grad():
            _, mask_u_w, _ = SF.pseudo_label(model_teacher(ecg_u_w, return_loss=False)['seg_logits'])
        model_student.train()
        logits = model_student(ops.batch_pair(ecg_x, ecg_u_w), return_loss=False)['seg_logits']
    return SF.fixmatch_loss(logits, ecg_x.size(0), mask_x, mask_u_w, None, 0.0)


def train_one_epoch(model_student: torch.nn.Module, model_teacher: torch.nn.Module, labeled_data_loader: Iterable,
                    unlabeled_data_loader: Iterable, optimizer: torch.optim.Optimizer, device: torch.device, epoch: int,
                    loss_scaler, log_writer=None, use_amp=True, config: Optional[dict] = None):
    """Self-training epoch with a frozen teacher; returns global averages of ``lr, loss_total, loss_x, loss_u_s``."""
    print_freq = 20
    accum_iter = config.get('accum_iter', 1)
    max_norm = config.get('max_norm', None)
    set_amp(use_amp, model_student)
    metric_logger = misc.MetricLogger(delimiter="  ")
    metric_logger.add_meter('lr', misc.SmoothedValue(window_size=1, fmt='{value:.6f}'))
    header = 'Epoch: [{}]'.format(epoch)
    model_student.train()
    model_teacher.eval()
    optimizer.zero_grad()
    num_steps = len(unlabeled_data_loader)
    assert len(labeled_data_loader) == num_steps, "The number of labeled and unlabeled data should be the same"
    buf = misc.DeviceMetricBuffer(['loss_total', 'loss_x', 'loss_u_s'], num_steps, device)
    lrs, logged = [], [0]

    def whole_step(ecg_x, mask_x, ecg_u_w):
        loss, stats = stpp_step(model_student, model_teacher, ecg_x, mask_x, ecg_u_w)
        loss_scaler(loss, optimizer, clip_grad=max_norm, parameters=model_student.parameters(), update_grad=True)
        optimizer.zero_grad()
        return stats[:3]

    # train.hip_graph (algorithms/base.py:step_graph_for); a new teacher (the next ST++ stage) re-captures
    graphed = step_graph_for(model_student, (id(model_teacher), id(optimizer), id(loss_scaler), max_norm, bool(use_amp)),
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
        ecg_u_w, _ = SA.unlabeled_views(unlabeled, device, want_strong=False)
        if graphed is not None:
            buf.push(graphed(ecg_x, mask_x, ecg_u_w))
        else:
            loss, stats = stpp_step(model_student, model_teacher, ecg_x, mask_x, ecg_u_w)
            buf.push(stats[:3])
            loss_scaler(loss / accum_iter if accum_iter != 1 else loss, optimizer, clip_grad=max_norm,
                        parameters=model_student.parameters(), update_grad=(data_iter_step + 1) % accum_iter == 0)
            if (data_iter_step + 1) % accum_iter == 0:
                optimizer.zero_grad()
        lr = max(g["lr"] for g in optimizer.param_groups)
        lrs.append(lr)
        metric_logger.update(lr=lr)
    flush()
    metric_logger.synchronize_between_processes()
    print('Averaged stats:', metric_logger)
    return {k: meter.global_avg for k, meter in metric_logger.meters.items()}


def _stage_dir(config, stage):
    return os.path.join(config['output_dir'], config['exp_name'], stage) if stage else \
        os.path.join(config['output_dir'], config['exp_name'])


def _stage_writer(config, stage):
    if misc.is_main_process() and config.get('output_dir'):
        from algorithms.base import make_log_writer
        d = _stage_dir(config, stage)
        os.makedirs(d, exist_ok=True)
        return d, make_log_writer(d)
    return None, None


def train_sup(config):
    """Stage 1 (``stpp.py:252-470``): labelled data only; checkpoints at epochs E//3, 2E//3, E."""
    device = setup_run(config)
    dist_on = config['ddp']['distributed']
    loader_train = get_dataloader(build_seg_dataset(config['dataset'], split='train_labeled'), is_distributed=dist_on,
                                  mode='train', **config['dataloader'])
    loader_valid = get_dataloader(build_seg_dataset(config['dataset'], split='valid'), is_distributed=dist_on,
                                  mode='valid', **config['dataloader'])
    output_dir, log_writer = _stage_writer(config, "stage1")
    model = build_model(config, device)
    print(f"Model = {model}")
    resolve_lr(config)
    model, model_without_ddp = wrap_ddp(config, model)
    optimizer = get_optimizer_from_config(config['train'], model_without_ddp.parameters())
    print(f"Optimizer = {optimizer}")
    loss_scaler = NativeScaler()
    best = {'loss': float('inf')}
    metric_fn = metrics_for(config)
    misc.load_model(config, model_without_ddp, optimizer, loss_scaler)
    num_epochs = config['train']['epochs']
    use_amp = config.get('use_amp', True)
    print(f"Start training for {num_epochs} epochs")
    start_time = time.time()
    for epoch in range(config['start_epoch'], num_epochs):
        if dist_on:
            loader_train.sampler.set_epoch(epoch)
        train_stats = train_one_epoch_labeled(model, loader_train, optimizer, device, epoch, loss_scaler, log_writer,
                                              use_amp=use_amp, config=config['train'])
        valid_stats, metrics, _, _ = evaluate(model, loader_valid, device, metric_fn, use_amp=use_amp, return_outputs=False)
        if output_dir and (epoch + 1) in [num_epochs // 3, num_epochs * 2 // 3, num_epochs]:
            misc.save_model(config, os.path.join(output_dir, f'checkpoint-{epoch + 1}.pth'), epoch, model_without_ddp,
                            optimizer, loss_scaler, metrics={'loss': valid_stats['loss'], **metrics})
        epoch_tail(config, output_dir, log_writer, epoch, model_without_ddp, optimizer, loss_scaler, train_stats,
                   valid_stats, metrics, best, metric_fn=metric_fn)
    print(f'Training time {datetime.timedelta(seconds=int(time.time() - start_time))}')
    if log_writer is not None:
        log_writer.close()
    _end_stage(model_without_ddp)


def _end_stage(*models):
    """The stage's HIP graph (train.hip_graph) goes with the stage: its pool, the tensors it keeps alive, and the model <-> graph
    reference cycle - freed here, not at some later cyclic-GC pass in the middle of the next stage."""
    import gc
    for m in models:
        drop_step_graph(m)
    gc.collect()


def prepare_semisup(config):
    """``stpp.py:473-509``: reload the three stage-1 checkpoints and rank the unlabelled records."""
    device = torch.device(config['device'])
    dataset = build_seg_dataset(config['dataset'], split='train_unlabeled', mode='eval')
    # the reference scores one record per forward (batch_size=1); the score is per record, so a full batch is equivalent
    loader = DataLoader(dataset, batch_size=config['dataloader'].get('reliability_batch_size', 64), shuffle=False,
                        num_workers=config['dataloader'].get('num_workers', 2),
                        pin_memory=config['dataloader'].get('pin_memory', False))
    models = []
    num_epochs = config['train']['epochs']
    for epoch in [num_epochs // 3, num_epochs * 2 // 3, num_epochs]:
        checkpoint = torch.load(os.path.join(_stage_dir(config, "stage1"), f'checkpoint-{epoch}.pth'),
                                map_location='cpu', weights_only=False)
        model = init_model_from_cfg(config)
        model.load_state_dict(checkpoint['model'])
        models.append(model.to(device))
    reliable_ids, _ = select_reliable(models, loader, device,
                                      reference_ids=config.get('stpp_reference_ids', False))
    return reliable_ids


def train_semisup(config, stage_id, unlabeled_subset_ids=None):
    """Stages 2 and 3 (``stpp.py:512-735``): teacher = best-<target_metric> checkpoint of the previous stage."""
    device = setup_run(config)
    ds_u = build_seg_dataset(config['dataset'], split='train_unlabeled')
    if unlabeled_subset_ids is not None:
        ds_u = torch.utils.data.Subset(ds_u, unlabeled_subset_ids)
    ds_l = build_seg_dataset(config['dataset'], split='train_labeled', num_unlabeled=len(ds_u))
    ds_v = build_seg_dataset(config['dataset'], split='valid')
    dist_on = config['ddp']['distributed']
    loader_l = get_dataloader(ds_l, is_distributed=dist_on, mode='train', **config['dataloader'])
    print(f"Labeled: {len(ds_l)} samples / {len(loader_l)} batches")
    loader_u = get_dataloader(ds_u, is_distributed=dist_on, mode='train', **config['dataloader'])
    print(f"Unlabeled: {len(ds_u)} samples / {len(loader_u)} batches")
    loader_v = get_dataloader(ds_v, is_distributed=dist_on, mode='valid', **config['dataloader'])
    output_dir, log_writer = _stage_writer(config, "stage2" if stage_id == 2 else None)
    model = build_model(config, device)
    print(f"Model = {model}")
    model_teacher = init_model_from_cfg(config).to(device)
    target_metric = config.get('test', {}).get('target_metric', "MeanIoU")
    teacher_path = os.path.join(_stage_dir(config, f"stage{stage_id - 1}"), f'best-{target_metric}.pth')
    print(f"Load teacher model from {teacher_path}")
    model_teacher.load_state_dict(torch.load(teacher_path, map_location='cpu', weights_only=False)['model'])
    for p in model_teacher.parameters():
        p.requires_grad = False
    resolve_lr(config)
    # the teacher only runs eval-mode forwards: no gradient to average, no batch statistics to sync -> not wrapped
    model, model_without_ddp = wrap_ddp(config, model)
    optimizer = get_optimizer_from_config(config['train'], model_without_ddp.parameters())
    print(f"Optimizer = {optimizer}")
    loss_scaler = NativeScaler()
    best = {'loss': float('inf')}
    metric_fn = metrics_for(config)
    misc.load_model(config, model_without_ddp, optimizer, loss_scaler)
    num_epochs = config['train']['epochs']
    use_amp = config.get('use_amp', True)
    print(f"Start training for {num_epochs} epochs")
    start_time = time.time()
    for epoch in range(config['start_epoch'], num_epochs):
        if dist_on:
            loader_l.sampler.set_epoch(epoch)
            loader_u.sampler.set_epoch(epoch)
        train_stats = train_one_epoch(model, model_teacher, loader_l, loader_u, optimizer, device, epoch, loss_scaler,
                                      log_writer, use_amp=use_amp, config=config['train'])
        valid_stats, metrics, _, _ = evaluate(model, loader_v, device, metric_fn, use_amp=use_amp, return_outputs=False)
        epoch_tail(config, output_dir, log_writer, epoch, model_without_ddp, optimizer, loss_scaler, train_stats,
                   valid_stats, metrics, best, metric_fn=metric_fn)
    print(f'Training time {datetime.timedelta(seconds=int(time.time() - start_time))}')
    if log_writer is not None:
        log_writer.close()
    _end_stage(model_without_ddp)


def _stage_barrier():
    if misc.is_dist_avail_and_initialized() and misc.get_world_size() > 1:
        torch.distributed.barrier()


def train(config):
    """``stpp.py:738-752``.  The process group is kept across the stages (the reference destroys it on the main
    process only, which would strand the other ranks)."""
    train_sup(config)
    _stage_barrier()   # rank 0 wrote checkpoint-*.pth / best-*.pth: nobody reads them before the writes are complete
    reliable_ids = prepare_semisup(config)
    train_semisup(config, stage_id=2, unlabeled_subset_ids=reliable_ids)
    _stage_barrier()
    train_semisup(config, stage_id=3)
