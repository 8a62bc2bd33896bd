"""Deterministic synthetic ECG windows, labels and model states (numpy only).

Everything the parity fixtures, the GPU tests, ``bench.py`` and the CPU
baseline feed to the hot path comes from this one counter-based generator, so
the GPU box (which never sees the reference) regenerates bit-identical inputs
and weights from a seed.  Shapes follow SURVEY.md §8(d):

* windows  ``ecg_* ~ N(0,1)`` fp32 ``(B, C, L)`` (the reference standardises
  every record, ``src/utils/transforms.py:290-310``),
* strong view ``ecg_u_s = ecg_u_w + 0.5 * N(0,1)`` (stand-in for RandAugment),
* delineation labels: piecewise-constant int64 in ``{0..3}``, runs of 50-200,
* weights: the reference's init law - conv ``N(0, 2/(k*Cout))`` and BN
  ``gamma=1, beta=0`` for the backbone (``src/models/backbones/resnet.py:326-333``),
  PyTorch's default ``U(-1/sqrt(fan_in), 1/sqrt(fan_in))`` for the FCN head.

The generator is splitmix64 over a 64-bit counter -> Box-Muller; no global
state, no dependence on numpy's or torch's RNG streams.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _key(seed: int, stream: int) -> np.uint64:
    k = _splitmix64(np.array([seed & 0xFFFFFFFFFFFFFFFF], dtype=np.uint64))
    k = _splitmix64(k ^ np.array([(stream * 0xD1342543DE82EF95) & 0xFFFFFFFFFFFFFFFF], dtype=np.uint64))
    return k[0]


def uniform(seed: int, stream: int, n: int, offset: int = 0) -> np.ndarray:
    """n doubles in (0, 1), element i depends only on (seed, stream, offset+i)."""
    with np.errstate(over="ignore"):
        ctr = np.arange(offset, offset + n, dtype=np.uint64) * np.uint64(0x2545F4914F6CDD1D) + _key(seed, stream)
    bits = _splitmix64(ctr) >> np.uint64(11)
    return (bits.astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def normal(seed: int, stream: int, shape, std: float = 1.0, mean: float = 0.0) -> np.ndarray:
    n = int(np.prod(shape))
    u1 = uniform(seed, 2 * stream, n)
    u2 = uniform(seed, 2 * stream + 1, n)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * math.pi * u2)
    return (mean + std * z).astype(np.float32).reshape(shape)


def uniform_pm(seed: int, stream: int, shape, bound: float) -> np.ndarray:
    n = int(np.prod(shape))
    return ((2.0 * uniform(seed, 2 * stream, n) - 1.0) * bound).astype(np.float32).reshape(shape)


def labels(seed: int, stream: int, B: int, L: int, num_classes: int = 4) -> np.ndarray:
    """Piecewise-constant int64 labels (B, L): runs of 50..200 samples."""
    out = np.empty((B, L), dtype=np.int64)
    max_runs = L // 50 + 2
    u_len = uniform(seed, 2 * stream, B * max_runs).reshape(B, max_runs)
    u_cls = uniform(seed, 2 * stream + 1, B * max_runs).reshape(B, max_runs)
    for b in range(B):
        pos, r = 0, 0
        while pos < L:
            run = 50 + int(u_len[b, r] * 151)
            out[b, pos:pos + run] = int(u_cls[b, r] * num_classes)
            pos += run
            r += 1
    return out


def fixmatch_batch(seed: int, B: int, C: int, L: int, num_classes: int = 4) -> dict:
    """One (labelled, unlabelled) pair of batches, keys as the reference's loaders
    yield them (``src/utils/semi_dataset.py:235-244``)."""
    ecg_x = normal(seed, 1, (B, C, L))
    ecg_u_w = normal(seed, 2, (B, C, L))
    ecg_u_s = ecg_u_w + 0.5 * normal(seed, 3, (B, C, L))
    return {
        "labeled": {"ecg": ecg_x, "target": labels(seed, 4, B, L, num_classes)},
        "unlabeled": {"ecg": ecg_u_w, "ecg_aug": ecg_u_s.astype(np.float32)},
    }


def learnable_batch(seed: int, B: int, C: int, L: int) -> dict:
    """A task the network can learn: the label (piecewise constant, runs of 50-200 samples) shifts the signal's local mean
    by (-1.5, -0.5, 0.5, 1.5) on every lead, plus N(0, 0.7) noise; the unlabelled windows are built the same way (their
    labels, ``u_target``, are never shown to a step), strong view = weak + N(0, 0.5)."""
    off = np.array([-1.5, -0.5, 0.5, 1.5], np.float32)
    yx, yu = labels(seed, 4, B, L), labels(seed, 5, B, L)
    x = (0.7 * normal(seed, 1, (B, C, L)) + off[yx][:, None, :]).astype(np.float32)
    uw = (0.7 * normal(seed, 2, (B, C, L)) + off[yu][:, None, :]).astype(np.float32)
    us = (uw + 0.5 * normal(seed, 3, (B, C, L))).astype(np.float32)
    return {"labeled": {"ecg": x, "target": yx}, "unlabeled": {"ecg": uw, "ecg_aug": us}, "u_target": yu}


# ---------------------------------------------------------------------------
# Model state (keys = the reference's state_dict keys)
# ---------------------------------------------------------------------------

def resnet18_fcn_spec(num_leads: int, num_classes: int = 4, head_channels: int = 128):
    """[(prefix, kind, shape_info)] in ``state_dict`` order for
    EncoderDecoder(resnet18-1D, FCNHead(num_convs=1, concat_input=False))."""
    spec = [("backbone.stem.0", "conv", (64, num_leads, 7)), ("backbone.stem.1", "bn", 64)]
    inpl = 64
    for li, planes in enumerate((64, 128, 256, 512), start=1):
        for bi in range(2):
            p = f"backbone.layer{li}.{bi}"
            cin = inpl if bi == 0 else planes
            spec += [(f"{p}.conv1", "conv", (planes, cin, 3)), (f"{p}.bn1", "bn", planes),
                     (f"{p}.conv2", "conv", (planes, planes, 3)), (f"{p}.bn2", "bn", planes)]
            if bi == 0 and (li > 1):
                spec += [(f"{p}.downsample.0", "conv", (planes, cin, 1)), (f"{p}.downsample.1", "bn", planes)]
        inpl = planes
    spec += [("decode_head.convs.0.0", "headconv", (head_channels, 512, 3)),
             ("decode_head.convs.0.1", "bn", head_channels),
             ("decode_head.cls_seg", "cls", (num_classes, head_channels, 1))]
    return spec


def model_state(seed: int, num_leads: int, num_classes: int = 4, trained: bool = False,
                sharpen: float = 1.0) -> "OrderedDict[str, np.ndarray]":
    """state_dict-shaped numpy arrays.

    ``trained=False``: the reference's init law.  ``trained=True``: perturbed BN
    affine parameters and running statistics (so eval-mode folding is exercised
    with non-trivial values) and cls weights scaled by ``sharpen`` (so that
    ``0 < mask_ratio < 1`` at ``conf_thresh`` 0.8, SURVEY.md §7 hard parts).
    """
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for i, (name, kind, info) in enumerate(resnet18_fcn_spec(num_leads, num_classes)):
        s = 100 + 8 * i
        if kind == "conv":
            co, ci, k = info
            sd[name + ".weight"] = normal(seed, s, info, std=math.sqrt(2.0 / (k * co)))
        elif kind == "headconv":
            co, ci, k = info
            sd[name + ".weight"] = uniform_pm(seed, s, info, 1.0 / math.sqrt(ci * k))
        elif kind == "cls":
            co, ci, k = info
            b = 1.0 / math.sqrt(ci * k)
            sd[name + ".weight"] = uniform_pm(seed, s, info, b) * np.float32(sharpen)
            sd[name + ".bias"] = uniform_pm(seed, s + 1, (co,), b) * np.float32(sharpen)
        else:
            c = info
            if trained:
                sd[name + ".weight"] = (1.0 + 0.2 * normal(seed, s, (c,))).astype(np.float32)
                sd[name + ".bias"] = (0.1 * normal(seed, s + 1, (c,))).astype(np.float32)
                sd[name + ".running_mean"] = (0.2 * normal(seed, s + 2, (c,))).astype(np.float32)
                sd[name + ".running_var"] = (1.0 + 0.5 * uniform(seed, s + 3, c)).astype(np.float32)
                sd[name + ".num_batches_tracked"] = np.array(7, dtype=np.int64)
            else:
                sd[name + ".weight"] = np.ones((c,), np.float32)
                sd[name + ".bias"] = np.zeros((c,), np.float32)
                sd[name + ".running_mean"] = np.zeros((c,), np.float32)
                sd[name + ".running_var"] = np.ones((c,), np.float32)
                sd[name + ".num_batches_tracked"] = np.array(0, dtype=np.int64)
    return sd


def param_keys(sd) -> list:
    """Keys of trainable parameters, in ``model.parameters()`` order."""
    return [k for k in sd if not (k.endswith("running_mean") or k.endswith("running_var")
                                  or k.endswith("num_batches_tracked"))]


def buffer_keys(sd) -> list:
    return [k for k in sd if k not in set(param_keys(sd))]


def checkpoint_family(seed_a, seed_b, num_leads, sharpen, count=3, spread=0.06):
    """``count`` model states that label records similarly but not identically (stand-ins for ST++'s three stage-1
    checkpoints): the state of ``seed_a`` moved ``spread * (count-1-k)/(count-1)`` of the way towards ``seed_b``; the
    last one is ``seed_a`` itself."""
    a = model_state(seed_a, num_leads, trained=True, sharpen=sharpen)
    b = model_state(seed_b, num_leads, trained=True, sharpen=sharpen)
    out = []
    for k in range(count):
        t = np.float32(spread * (count - 1 - k) / max(count - 1, 1))
        out.append({key: (v if v.dtype.kind != "f" else (v + t * (b[key] - v)).astype(np.float32)) for key, v in a.items()})
    return out
