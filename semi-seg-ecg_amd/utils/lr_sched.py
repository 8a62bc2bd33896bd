"""Per-iteration learning-rate schedule (``src/utils/lr_sched.py:6-18``): linear warm-up
then half-cycle cosine to ``min_lr``; honours a per-group ``lr_scale``.  Host-side floats only."""
import math


def lr_at(epoch, config):
    if epoch < config['warmup_epochs']:
        return config['lr'] * epoch / config['warmup_epochs']
    span = config['epochs'] - config['warmup_epochs']
    return config['min_lr'] + (config['lr'] - config['min_lr']) * 0.5 * \
        (1. + math.cos(math.pi * (epoch - config['warmup_epochs']) / span))


def adjust_learning_rate(optimizer, epoch, config):
    lr = lr_at(epoch, config)
    for param_group in optimizer.param_groups:
        param_group["lr"] = lr * param_group["lr_scale"] if "lr_scale" in param_group else lr
    return lr
