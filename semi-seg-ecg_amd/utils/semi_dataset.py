"""Loader boundary of the hot path.

The reference's record pipeline (``src/utils/semi_dataset.py`` + ``transforms.py``: pickled
records, Butterworth filtering, FFT resize-crop, RandAugment) is host-side numpy/scipy code
that SURVEY.md §8 keeps OUT of the MI355X hot path; it is not re-implemented here.  This
module provides (a) the same two entry points the algorithm plugins call
(``build_seg_dataset``, ``get_dataloader``), (b) a deterministic synthetic dataset with the
reference's batch-dict keys (``ecg (C,L) f32``, ``target (L,) i64``, ``ecg_aug (C,L) f32``),
selected with ``dataset: {synthetic: {...}}``, and (c) two hooks to the reference's OWN record
pipeline, used where it lies (nothing is copied): ``dataset.reference_src: /path/to/reference/src``
loads its ``utils/transforms.py`` + ``utils/semi_dataset.py`` under private module names and calls
its ``build_seg_dataset``; ``dataset.builder: "package.module:function"`` delegates to any
importable builder with that signature.
"""
from __future__ import annotations

import importlib

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset, DistributedSampler, RandomSampler, SequentialSampler

from ssecg import synth


class SyntheticECGDataset(Dataset):
    def __init__(self, num_samples, num_leads, signal_length, split, seed=1234, num_classes=4, raw_unlabeled=False):
        self.n, self.C, self.L, self.split, self.seed, self.K = num_samples, num_leads, signal_length, split, seed, num_classes
        self.raw_unlabeled = raw_unlabeled

    def __len__(self):
        return self.n

    def __getitem__(self, idx):
        s = self.seed * 1000003 + idx
        ecg = synth.normal(s, 1, (self.C, self.L))
        if self.split == "train_unlabeled" and self.raw_unlabeled:
            # device pipeline (dataset.device_augment): the loader stops after filter/crop and hands over the raw record
            return {"ecg_raw": torch.from_numpy((0.2 + 1.5 * ecg).astype(np.float32))}
        item = {"ecg": torch.from_numpy(ecg)}
        if self.split == "train_unlabeled":
            item["ecg_aug"] = torch.from_numpy((ecg + 0.5 * synth.normal(s, 3, (self.C, self.L))).astype(np.float32))
        else:
            item["target"] = torch.from_numpy(synth.labels(s, 4, 1, self.L, self.K)[0])
        return item


_REF_MODULES = {}


def load_reference_pipeline(src_dir: str):
    """The reference's ``utils.semi_dataset`` module, loaded from ``<src_dir>/utils/`` WITHOUT putting the reference on
    ``sys.path``: this source root carries the reference's package names on purpose (``utils``, ``algorithms``, ``models`` - it is
    a drop-in), so ``import utils.semi_dataset`` can only ever mean one of the two.  The reference's two files are executed
    under private names while ``utils`` / ``utils.transforms`` / ``utils.misc`` are mapped, for the duration of the load only,
    to a stand-in package that resolves ``utils.transforms`` to the reference's file and ``utils.misc`` to THIS package's
    ``utils.misc`` (same ``get_rank`` / ``get_world_size``; the reference's own ``utils/misc.py`` imports ``torch._six``, which
    no torch >= 2 has: SURVEY Q1).  Names bound at import time keep pointing at the reference's objects afterwards."""
    import importlib.util
    import os
    import sys
    import types
    src_dir = os.path.realpath(src_dir)
    if src_dir in _REF_MODULES:
        return _REF_MODULES[src_dir]
    udir = os.path.join(src_dir, "utils")
    for f in ("transforms.py", "semi_dataset.py"):
        if not os.path.exists(os.path.join(udir, f)):
            raise FileNotFoundError(f"dataset.reference_src: {os.path.join(udir, f)} not found")
    import utils.misc as own_misc
    saved = {k: sys.modules.get(k) for k in ("utils", "utils.transforms", "utils.misc", "utils.semi_dataset")}
    try:
        def load(private_name, path):
            spec = importlib.util.spec_from_file_location(private_name, path)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            return mod
        tr = load("_ssecg_reference_transforms", os.path.join(udir, "transforms.py"))
        pkg = types.ModuleType("utils")
        pkg.__path__ = []            # a package whose submodules are exactly the two entries below
        pkg.transforms, pkg.misc = tr, own_misc
        sys.modules["utils"], sys.modules["utils.transforms"], sys.modules["utils.misc"] = pkg, tr, own_misc
        sd = load("_ssecg_reference_semi_dataset", os.path.join(udir, "semi_dataset.py"))
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    _REF_MODULES[src_dir] = sd
    return sd


def build_seg_dataset(cfg: dict, split: str, num_unlabeled=None, **kwargs):
    if cfg.get("reference_src"):
        ref_cfg = {k: v for k, v in cfg.items() if k not in ("reference_src", "builder", "synthetic", "device_augment")}
        return load_reference_pipeline(cfg["reference_src"]).build_seg_dataset(ref_cfg, split=split, num_unlabeled=num_unlabeled,
                                                                               **kwargs)
    if cfg.get("builder"):
        mod, fn = cfg["builder"].split(":")
        return getattr(importlib.import_module(mod), fn)(cfg, split=split, num_unlabeled=num_unlabeled, **kwargs)
    syn = cfg.get("synthetic")
    if syn is None:
        raise NotImplementedError(
            "record datasets stay on the reference's host pipeline (outside the MI355X hot path): set "
            "`dataset.reference_src: /path/to/reference/src` to use it where it lies, `dataset.builder: 'pkg.mod:fn'` for "
            "any other builder, or `dataset.synthetic` for generated windows (INTEGRATION.md)")
    n = {"train_unlabeled": syn.get("num_unlabeled", 256), "train_labeled": syn.get("num_labeled", 64),
         "valid": syn.get("num_valid", 64), "test": syn.get("num_test", 64)}[split]
    if split == "train_labeled" and num_unlabeled is not None:
        n = num_unlabeled  # the reference over-samples the labelled set to the unlabelled length (:86-95)
    seed = syn.get("seed", 1234) + {"train_unlabeled": 0, "train_labeled": 1, "valid": 2, "test": 3}[split]
    return SyntheticECGDataset(n, syn.get("num_leads", 1), cfg.get("signal_length", 2000), split, seed,
                               syn.get("num_classes", 4), raw_unlabeled=bool(cfg.get("device_augment")))


def get_dataloader(dataset, is_distributed=False, dist_eval=False, mode="train", **kwargs):
    """Sampler policy of ``src/utils/semi_dataset.py:325-362``: shuffled (distributed) sampler and
    ``drop_last`` for training, sequential otherwise."""
    train = mode == "train"
    kwargs = dict(kwargs)
    drop_last = kwargs.pop("drop_last", None)     # src/utils/semi_dataset.py:354-356: an explicit value wins, else = train
    drop_last = train if drop_last is None else bool(drop_last)
    kwargs.pop("reliability_batch_size", None)    # ST++'s scoring batch (algorithms/stpp.py:prepare_semisup), not a DataLoader key
    if is_distributed and (train or dist_eval):
        sampler = DistributedSampler(dataset, shuffle=train)
    else:
        sampler = RandomSampler(dataset) if train else SequentialSampler(dataset)
    return DataLoader(dataset, sampler=sampler, drop_last=drop_last, **kwargs)


# ----------------------------------------------------------------------------------------------------------------------
# Host -> device hand-over.  The reference moves every batch with ``.to(device, non_blocking=True)`` at the top of the step
# (src/algorithms/fixmatch.py:80-84): on the compute stream that copy is serial with the kernels - at B = 512, 12 leads a step
# consumes 156 MB of windows and labels = 3.05 ms of a 23.0 ms (fp32) / 10.5 ms (bf16) step (tools/pcie_step_bench.py).
class DevicePrefetcher:
    """Iterate ``loader`` with batch i + 1 already travelling to ``device`` on a side HIP stream while step i computes.

    Yields the loader's own batch structure (dict / list / tuple of tensors and anything else) with every tensor on the
    device; the plugins' own ``.to(device, non_blocking=True)`` calls then are no-ops.  ``len()`` and sampler access go to the
    wrapped loader.  Batches whose tensors already live on the device (tests, the bench) pass through untouched.  Measured
    (B = 512, C = 12, pinned host batches): fp32 26.08 -> 23.57 ms/step (HBM-resident 23.03), bf16 13.50 -> 10.96 (10.47)."""

    def __init__(self, loader, device):
        self.loader = loader
        self.device = torch.device(device)
        self._stream = None

    def __len__(self):
        return len(self.loader)

    def __getattr__(self, name):        # .sampler, .dataset, .batch_size ... of the wrapped loader
        if name in ("loader", "device", "_stream"):     # not set yet (copy / unpickle): no recursion through self.loader
            raise AttributeError(name)
        return getattr(self.loader, name)

    def _move(self, obj, moved):
        if isinstance(obj, torch.Tensor):
            if obj.device == self.device:
                return obj
            out = obj.to(self.device, non_blocking=True)
            moved.append(out)
            return out
        if isinstance(obj, dict):
            return {k: self._move(v, moved) for k, v in obj.items()}
        if isinstance(obj, (list, tuple)):
            return type(obj)(self._move(v, moved) for v in obj)
        return obj

    def __iter__(self):
        if self.device.type != "cuda":
            yield from self.loader
            return
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=self.device)
        side = self._stream
        it = iter(self.loader)

        def fetch():
            try:
                batch = next(it)
            except StopIteration:
                return None
            moved = []
            with torch.cuda.stream(side):
                batch = self._move(batch, moved)
            return batch, moved

        nxt = fetch()
        while nxt is not None:
            batch, moved = nxt
            cur = torch.cuda.current_stream(self.device)
            if moved:
                cur.wait_stream(side)                 # the copies of THIS batch are done before the step's first kernel
                for t in moved:
                    t.record_stream(cur)              # allocated on the side stream, consumed on the compute stream
            nxt = fetch()                             # batch i + 1 starts travelling now, under step i
            yield batch


def device_prefetch(loader, device, enabled=True):
    """-> ``DevicePrefetcher(loader, device)`` on a HIP device (``train.device_prefetch: false`` in the YAML turns it off)."""
    if not enabled or torch.device(device).type != "cuda" or isinstance(loader, DevicePrefetcher):
        return loader
    return DevicePrefetcher(loader, device)

