"""Optimizer factory (``src/utils/optimizer.py:8-37``): ``adamw`` and ``sgd``, both fused multi-tensor HIP updates
with ``torch.optim``-compatible ``state_dict`` layouts."""
from ssecg.optim import FusedAdamW, FusedSGD


def get_optimizer_from_config(config: dict, param_groups):
    opt_name = config['optimizer']
    lr = config['lr']
    weight_decay = config['weight_decay']
    kwargs = config.get('optimizer_kwargs', {}) or {}
    if opt_name == "sgd":
        return FusedSGD(param_groups, lr=lr, momentum=kwargs.get('momentum', 0), weight_decay=weight_decay)
    if opt_name == "adamw":
        betas = kwargs.get('betas', (0.9, 0.999))
        if isinstance(betas, list):
            betas = tuple(betas)
        return FusedAdamW(param_groups, lr=lr, betas=betas, eps=kwargs.get('eps', 1e-8), weight_decay=weight_decay)
    raise ValueError(f"Unknown optimizer: {opt_name}")
