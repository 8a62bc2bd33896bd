"""TEST INFRASTRUCTURE ONLY - numpy restatement of the segmentation metrics on the validation / ST++ path.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

* ``mean_iou_update`` / ``MeanIoURef`` restate torchmetrics==1.5.2 (requirements.txt:12 of the reference; the
  package is NOT installed in this image, so this part is PARITY UNPINNED: it follows the published algorithm of
  torchmetrics/functional/segmentation/mean_iou.py::_mean_iou_update/_mean_iou_compute and
  torchmetrics/segmentation/mean_iou.py::MeanIoU.update/compute, as called from src/algorithms/base.py:206-231
  with the kwargs of src/utils/perf_metrics.py:9-25).
* ``calculate_miou`` / ``select_reliable`` restate src/algorithms/stpp.py:32-42 and :45-88 and ARE pinned against
  the imported reference (tools/make_golden.py -> tests/golden/stpp_select.npz).
"""
import numpy as np


def one_hot(idx, K):
    """(N, L) int -> (N, K, L) bool, as F.one_hot(..).movedim(-1, 1) (src/algorithms/base.py:208-216)."""
    return (np.arange(K)[None, :, None] == idx[:, None, :])


def mean_iou_update(preds, target, include_background=True):
    """preds/target: one-hot (N, K, L) bool -> intersection, union (N, K') int64."""
    if not include_background:
        preds, target = preds[:, 1:], target[:, 1:]
    inter = (preds & target).sum(axis=2).astype(np.int64)
    union = target.sum(axis=2).astype(np.int64) + preds.sum(axis=2).astype(np.int64) - inter
    return inter, union


def mean_iou_compute(inter, union, per_class=False):
    val = np.where(union != 0, inter / np.where(union != 0, union, 1), 0.0)  # _safe_divide(.., zero_division=0)
    return val if per_class else val.mean(axis=1)


class MeanIoURef:
    def __init__(self, num_classes, include_background=True, per_class=False):
        self.K, self.bg, self.per_class = num_classes, include_background, per_class
        self.score, self.num_batches = 0.0, 0

    def update(self, pred_idx, target_idx):
        i, u = mean_iou_update(one_hot(pred_idx, self.K), one_hot(target_idx, self.K), self.bg)
        s = mean_iou_compute(i, u, self.per_class)
        self.score = self.score + (s.mean(axis=0) if self.per_class else s.mean())
        self.num_batches += 1

    def compute(self):
        return self.score / self.num_batches


def calculate_miou(onehot_preds, onehot_labels, ignore_background=False):
    """src/algorithms/stpp.py:32-42 - pooled over the whole array (all records passed in), float one-hots."""
    if ignore_background:
        onehot_preds, onehot_labels = onehot_preds[:, 1:], onehot_labels[:, 1:]
    ious = []
    for i in range(onehot_preds.shape[1]):
        inter = (onehot_preds[:, i] * onehot_labels[:, i]).sum()
        union = onehot_preds[:, i].sum() + onehot_labels[:, i].sum() - inter
        ious.append(inter / union if union > 0 else 0.0)
    return np.mean(ious)


def reliabilities(pred_idx_per_model, K):
    """Per-record reliability of src/algorithms/stpp.py:63-80: mean over the earlier checkpoints of the mIoU between
    their prediction and the LAST checkpoint's.  pred_idx_per_model: list of (R, L) int arrays -> (R,) float64."""
    R = pred_idx_per_model[0].shape[0]
    out = np.zeros(R)
    for r in range(R):
        oh = [one_hot(p[r:r + 1], K).astype(np.int64) for p in pred_idx_per_model]
        m = [calculate_miou(oh[i], oh[-1]) for i in range(len(oh) - 1)]
        out[r] = sum(m) / len(m)
    return out


def select_reliable_ids(rel, num_models=None, reference_ids=False):
    """src/algorithms/stpp.py:82-88: stable descending sort, first half reliable.  ``reference_ids=True`` reproduces
    the reference's returned ids literally: its inner ``for i in range(len(onehot_preds) - 1)`` loop (stpp.py:72)
    overwrites the record index ``i`` before ``id_to_reliability.append((i, reliability))`` (stpp.py:81), so every
    entry carries the id ``num_models - 2``."""
    ids = [num_models - 2] * len(rel) if reference_ids else list(range(len(rel)))
    pairs = sorted(zip(ids, rel), key=lambda e: e[1], reverse=True)
    half = len(pairs) // 2
    return [e[0] for e in pairs[:half]], [e[0] for e in pairs[half:]]
