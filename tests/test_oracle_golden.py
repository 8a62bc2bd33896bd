"""CPU: the oracle (oracle/torch_ref.py) re-checked against the reference outputs frozen in tests/golden/
(the reference itself never leaves the build container).  The oracle is the checker used by every GPU test,
bench.py's cpu_baseline and smoke(); this file is what pins it."""
from collections import OrderedDict

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import DROPOUT_P, L, TRAIN_CFG, check_packed, dropout_mask_np, golden, sharpen_for
from oracle import torch_ref as O
from ssecg import synth


def _t(batch):
    return {g: {k: torch.from_numpy(v) for k, v in d.items()} for g, d in batch.items()}


@pytest.mark.parametrize("C,B,seed", [(1, 2, 11), (2, 2, 12), (12, 2, 13)])
def test_oracle_forward_matches_reference(C, B, seed):
    g = golden(f"forward_c{C}_b{B}")
    sd = O.state_from_numpy(synth.model_state(seed, C, trained=True, sharpen=sharpen_for(C)))
    x = torch.from_numpy(synth.normal(seed + 1, 1, (B, C, L)))
    with torch.no_grad():
        logits = O.model_forward(sd, x, train=False)
        conf, mask = O.pseudo_label(logits)
    assert (logits - torch.from_numpy(g["eval.logits"])).abs().max().item() < 1e-5
    assert np.array_equal(mask.numpy().astype(np.int8), g["eval.mask"])
    assert np.array_equal((conf >= TRAIN_CFG["conf_thresh"]).numpy(), g["eval.keep"])
    y = torch.from_numpy(synth.labels(seed + 1, 4, B, L))
    dm = torch.from_numpy(dropout_mask_np(seed + 1, B).astype(np.float32))
    lt = O.model_forward(sd, x, train=True, dropout_mask=dm, dropout_p=DROPOUT_P)
    assert (lt.detach() - torch.from_numpy(g["train.logits"])).abs().max().item() < 1e-5
    loss = F.cross_entropy(lt, y)
    assert abs(loss.item() - float(g["train.loss"])) < 1e-6
    names = O.param_names(sd)
    grads = dict(zip(names, torch.autograd.grad(loss, [sd[k] for k in names])))
    check_packed(g, "train.grad.", grads, 1e-5, what="oracle gradients")
    check_packed(g, "train.buf.", {k: sd[k] for k in O.buffer_names(sd)}, 1e-6, what="oracle buffers")


@pytest.mark.parametrize("algo,C,B,seed", [("fixmatch", 1, 2, 21), ("mean_teacher", 2, 2, 23), ("base", 1, 2, 24)])
def test_oracle_steps_match_reference(algo, C, B, seed):
    g = golden(f"{algo}_c{C}_b{B}")
    sd = O.state_from_numpy(synth.model_state(seed, C, trained=True, sharpen=sharpen_for(C)))
    teacher = None
    if algo == "mean_teacher":  # teacher parameters alias the student's at construction (Q4); own buffers
        tb = O.state_from_numpy(synth.model_state(seed + 50, C, trained=True, sharpen=sharpen_for(C)), requires_grad=False)
        pn = set(O.param_names(sd))
        teacher = OrderedDict((k, sd[k] if k in pn else tb[k]) for k in sd)
    opt, cfg = {}, dict(TRAIN_CFG, betas=(0.9, 0.999))
    for s in range(2):
        epoch = 3 + 9 * s
        batch = _t(synth.fixmatch_batch(seed + 10 + s, B, C, L))
        pre = f"step{s}."
        n = B if algo == "base" else 2 * B
        dm = torch.from_numpy(dropout_mask_np(seed + 10 + s, n).astype(np.float32))
        if algo == "base":
            r = O.supervised_step(sd, opt, batch["labeled"], cfg, epoch, dm)
            assert abs(r["loss"] - float(g[pre + "loss"])) < 1e-6
        elif algo == "fixmatch":
            r = O.fixmatch_step(sd, opt, batch, cfg, epoch, dm)
            assert np.array_equal(r["mask"].numpy().astype(np.int8), g[pre + "mask"])
            assert np.array_equal(r["keep"].numpy(), g[pre + "keep"])
            assert abs(r["mask_ratio"] - float(g[pre + "mask_ratio"])) < 1e-6
        else:
            r = O.mean_teacher_step(sd, teacher, opt, batch, cfg, epoch, dm)
        if algo != "base":
            for k in ("loss_total", "loss_x", "loss_u_s"):
                assert abs(r[k] - float(g[pre + k])) < 1e-6
        assert abs(r["lr"] - float(g[pre + "lr"])) < 1e-15
        assert (r["logits"] - torch.from_numpy(g[pre + "logits"])).abs().max().item() < 2e-5
        check_packed(g, pre + "param.", {k: sd[k] for k in O.param_names(sd)}, 1e-6, what="oracle params")
        if teacher is not None:
            check_packed(g, pre + "tparam.", {k: teacher[k] for k in O.param_names(sd)}, 1e-6, what="oracle teacher")
            assert str(teacher["backbone.stem.1.num_batches_tracked"].dtype) == str(g[pre + "tbuf.nbt_dtype"])
