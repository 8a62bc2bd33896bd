"""Validation metrics (``src/utils/perf_metrics.py``): ``build_metric_fn`` / ``is_best_metric`` with the same
signatures, backed by per-record confusion counts computed on the MI355X (``ssecg_seg_confusion``) instead of
torchmetrics fed with all-gathered (B, K, L) one-hot tensors on the CPU.

The reference pins ``torchmetrics==1.5.2`` (requirements.txt:12), which is not installed here; ``MeanIoU`` below
restates its published algorithm (torchmetrics/segmentation/mean_iou.py + functional/segmentation/mean_iou.py @1.5.2):
  update : intersection[n,c] = sum(pred & target), union[n,c] = sum(pred) + sum(target) - intersection  (per record);
           iou = intersection / union with 0 where union == 0; score += mean_c(iou).mean_n()  (or mean_n per class);
           num_batches += 1
  compute: score / num_batches
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch

from ssecg import functional as SF


class MeanIoU:
    higher_is_better = True

    def __init__(self, num_classes: int, include_background: bool = True, per_class: bool = False,
                 input_format: str = "one-hot", **_unused):
        if input_format not in ("one-hot", "index"):
            raise ValueError(f"Expected argument `input_format` to be one of 'one-hot', 'index', but got {input_format}.")
        if not isinstance(num_classes, int) or num_classes <= 0:
            raise ValueError(f"Expected argument `num_classes` must be a positive integer, but got {num_classes}.")
        self.num_classes = num_classes
        self.include_background = include_background
        self.per_class = per_class
        self.input_format = input_format
        self.reset()

    def reset(self):
        self.score = None
        self.num_batches = 0

    def to(self, device):
        return self

    def update_counts(self, counts: torch.Tensor) -> None:
        """counts: (N, K, K) per-record confusion counts of ONE batch (all ranks' records, like the reference's
        all-gathered batch)."""
        iou = SF.iou_from_confusion(counts, self.include_background)
        score = iou.mean(dim=0) if self.per_class else iou.mean(dim=1).mean()
        self.score = score if self.score is None else self.score + score
        self.num_batches += 1

    def update(self, preds: torch.Tensor, target: torch.Tensor) -> None:
        """torchmetrics calling convention: one-hot (N, K, L) or index (N, L) tensors."""
        if self.input_format == "one-hot":
            if preds.shape != target.shape or preds.shape[1] != self.num_classes:
                raise RuntimeError("Predictions and targets are expected to have the same shape (N, num_classes, ...)")
            preds, target = preds.argmax(dim=1), target.argmax(dim=1)
        self.update_counts(SF.seg_confusion(preds.reshape(preds.shape[0], -1), target.reshape(target.shape[0], -1),
                                            self.num_classes))

    def compute(self) -> torch.Tensor:
        if self.score is None:
            raise RuntimeError("MeanIoU.compute() called before update()")
        return (self.score / self.num_batches).to(torch.float32)


class MetricCollection(dict):
    """name -> metric (the subset of torchmetrics.MetricCollection the reference uses)."""

    def __init__(self, metrics):
        super().__init__({m.__class__.__name__: m for m in metrics})

    def to(self, device):
        return self

    def update(self, *args, **kwargs):  # noqa: D102  (shadows dict.update on purpose, like torchmetrics)
        for m in self.values():
            m.update(*args, **kwargs)

    def update_counts(self, counts):
        for m in self.values():
            m.update_counts(counts)

    def compute(self):
        return {k: m.compute() for k, m in self.items()}

    def reset(self):
        for m in self.values():
            m.reset()


_METRICS = {"MeanIoU": MeanIoU}


def build_metric_fn(config: dict) -> Tuple[MetricCollection, Dict[str, float]]:
    """``src/utils/perf_metrics.py:9-47``."""
    common = {}
    if config["task"] == "segmentation":
        common["num_classes"] = config["num_classes"]
        common["include_background"] = config.get("include_background", True)
        common["per_class"] = config.get("per_class", False)
        common["input_format"] = config.get("input_format", "one-hot")
    else:
        raise ValueError(f"Invalid task: {config['task']}")
    metric_list = []
    for name in config["target_metrics"]:
        kwargs = dict(common)
        if isinstance(name, dict):
            assert len(name) == 1, f"Invalid metric name: {name}"
            name, extra = list(name.items())[0]
            kwargs = {**extra, **common}
        assert isinstance(name, str), f"metric name must be a string: {name}"
        assert name in _METRICS, f"Invalid metric name: {name}"
        metric_list.append(_METRICS[name](**kwargs))
    metric_fn = MetricCollection(metric_list)
    best = {k: -float("inf") if v.higher_is_better else float("inf") for k, v in metric_fn.items()}
    return metric_fn, best


def is_best_metric(metric_class, prev_metric: float, curr_metric: float) -> bool:
    """``src/utils/perf_metrics.py:50-60``."""
    return curr_metric > prev_metric if metric_class.higher_is_better else curr_metric < prev_metric
