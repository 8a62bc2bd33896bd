"""Shared test plumbing: build the HIP model / oracle state from ssecg.synth, the fixed dropout masks
the golden fixtures were generated with, and tolerance helpers."""
import os

import numpy as np
import torch

from ssecg import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
L = 2000
DROPOUT_P = 0.1
TRAIN_CFG = dict(epochs=100, accum_iter=1, warmup_epochs=10, min_lr=1e-4, lr=1e-3, weight_decay=0.05, max_norm=None,
                 optimizer="adamw", optimizer_kwargs={"betas": [0.9, 0.999]}, conf_thresh=0.80, ema_decay=0.99)


def sharpen_for(C):
    return {1: 5.0, 2: 16.0, 12: 24.0}[C]


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def dropout_mask_np(seed, n, lp=63, ch=128, p=DROPOUT_P):
    u = synth.uniform(seed, 77, n * ch * lp).reshape(n, ch, lp)
    return (u >= p)


def model_cfg(C, dropout_ratio=DROPOUT_P):
    return {"backbone": {"resnet18": dict(num_leads=C, num_stages=4, out_indices=[0, 1, 2, 3], dilations=[1, 1, 1, 1],
                                           strides=[1, 2, 2, 2], deep_stem=False, avg_down=False, contract_dilation=False)},
            "decode_head": {"FCNHead": dict(in_channels=512, in_index=3, channels=128, num_convs=1, concat_input=False,
                                            dropout_ratio=dropout_ratio, num_classes=4, align_corners=False)}}


def build_hip_model(C, sd_np, device):
    from algorithms.base import init_model_from_cfg
    model = init_model_from_cfg(model_cfg(C))
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()})
    return model.to(device)


def to_dev(batch, device):
    return {g: {k: torch.from_numpy(v).to(device) for k, v in d.items()} for g, d in batch.items()}


def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def tstats(t):
    t = t.detach().double().cpu()
    return np.array([t.sum().item(), t.abs().sum().item(), t.pow(2).sum().sqrt().item()])


def check_packed(g, prefix, named, tol, atol_full=None, what="", flip_tolerant=False):
    """Compare a dict of tensors with a pack_tensors() record of the golden file.

    Fixtures record (``<prefix>noise``) how far the fp32 REFERENCE itself sits from an fp64 evaluation of the
    same graph: where a ReLU / max-pool / threshold decision is a near-tie the reference's own gradients move by
    1e-3, elsewhere by 1e-6.  The bar is the nominal tolerance, widened to a few times that measured floor.

    ``flip_tolerant`` (model-level GRADIENT checks): with B=2 windows a single ReLU decision that differs between
    two correct fp32 implementations (pre-activation within rounding noise of 0; ~0.5 expected per fixture) shifts
    whole gradient tensors by O(0.5%).  Kernel-level gradient parity is pinned tightly (2e-5) by tests/test_ops_gpu.py,
    where the ReLU mask is a shared input; here the bar is checksum 1e-2 / relative L2 2e-2.

    ``atol_full`` selects the post-optimizer mode: AdamW moves every weight by ~lr*sign-like steps, so an element
    whose gradient is numerically ~0 may step the other way; full tensors are then judged by max |d| <= atol_full
    and RMS(d) <= 0.2*atol_full (a wrong update direction would give RMS ~ 0.6*atol_full)."""
    names = [str(n) for n in g[prefix + "names"]]
    stats = g[prefix + "stats"]
    assert set(names) == set(named.keys()), f"{what}: tensor name sets differ"
    nz = float(g[prefix + "noise"].max()) if (prefix + "noise") in g.files else 0.0
    tol_sum = max(tol, 4.0 * nz, 1e-2 if flip_tolerant else 0.0)
    tol_full = max(5.0 * tol, 40.0 * nz)
    worst = 0.0
    for i, k in enumerate(names):
        s = tstats(named[k])
        ref = stats[i]
        # L2 norm and abs-sum are robust summaries; the plain sum is compared on the abs-sum scale
        e = max(abs(s[2] - ref[2]) / (ref[2] + 1e-12), abs(s[1] - ref[1]) / (ref[1] + 1e-12),
                abs(s[0] - ref[0]) / (ref[1] + 1e-12))
        worst = max(worst, e)
        assert e < tol_sum, f"{what} {k}: checksum rel err {e:.3e} >= {tol_sum:.1e}"
        fk, sk = prefix + "full." + k, prefix + "slice." + k
        if fk in g.files:
            ref_t = torch.from_numpy(g[fk])
            got = named[k].detach().float().cpu()
            if atol_full is None and not flip_tolerant:
                assert rel(got, ref_t) < tol_full, f"{what} {k}: full tensor rel err {rel(got, ref_t):.3e} >= {tol_full:.1e}"
            elif atol_full is None:
                # one flipped ReLU decision (an activation within fp32 noise of 0) moves ONE channel's sum by O(10%)
                # and everything upstream by O(0.5%): judge the bulk of the tensor, not its worst element
                l2 = ((got - ref_t).double().norm() / (ref_t.double().norm() + 1e-30)).item()
                assert l2 <= 2e-2, f"{what} {k}: relative L2 error {l2:.2e}"
                bad = ((got - ref_t).abs() > 2e-2 * ref_t.abs().max()).float().mean().item()
                assert bad <= max(0.02, 2.0 / ref_t.numel()), f"{what} {k}: {bad:.1%} of elements off by > 2%"
            else:
                d = (got - ref_t).abs()
                assert d.max().item() <= atol_full, f"{what} {k}: max |d| {d.max().item():.3e} > {atol_full}"
                rms = d.pow(2).mean().sqrt().item()
                assert rms <= 0.2 * atol_full, f"{what} {k}: RMS(d) {rms:.3e} > {0.2 * atol_full:.1e}"
        if sk in g.files:
            ref_t = torch.from_numpy(g[sk])
            got = named[k].detach().float().cpu()[:8, :8]
            if atol_full is None and flip_tolerant:
                l2 = ((got - ref_t).double().norm() / (ref_t.double().norm() + 1e-30)).item()
                assert l2 <= 5e-2, f"{what} {k}: slice relative L2 error {l2:.2e}"
            elif atol_full is None:
                rms_ref = float(ref[2]) / np.sqrt(named[k].numel()) + 1e-12
                assert ((got - ref_t).abs().max() / rms_ref).item() < 10 * tol_full, f"{what} {k}: slice mismatch"
            else:
                assert (got - ref_t).abs().max().item() <= atol_full
    return worst


# ---- re-anchoring: the oracle continued from the DEVICE's own state ---------------------------------------------------
# Comparing a second optimisation step against a fixture is ill-conditioned: AdamW's first update is lr*g/(|g|+eps), i.e.
# sign-like, so weights whose gradient sits at the rounding-noise level land +-lr apart for ANY two fp32 implementations
# and the step-1 logits inherit that (measured 3e-3..6e-3 on the sharpened fixtures).  The chain used instead is
#   reference == oracle for two steps (bit-identical, tests/test_oracle_golden.py, CPU)
#   device step 0 == reference step 0 (1e-4, golden fixtures)
#   device step 1 == oracle step 1 started from the device's post-step-0 weights / moments / buffers (1e-4).
def oracle_state(model, requires_grad=True):
    from oracle import torch_ref as O
    return O.state_from_numpy({k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}, requires_grad=requires_grad)


def oracle_opt(optimizer, model):
    opt = {}
    for k, p in model.named_parameters():
        st = optimizer.state.get(p)
        if st:
            opt["step"] = int(torch.as_tensor(st["step"]).item())
            opt["exp_avg." + k] = st["exp_avg"].detach().cpu().clone()
            opt["exp_avg_sq." + k] = st["exp_avg_sq"].detach().cpu().clone()
    return opt


def cpu_batch(batch_np):
    return {g: {k: torch.from_numpy(v) for k, v in d.items()} for g, d in batch_np.items()}


def check_params_vs_oracle(model, o_sd, lr, what=""):
    """Parameters after an AdamW step vs the oracle's: every element within ~2 updates (2.2*lr), RMS within 0.1*lr."""
    sd = model.state_dict()
    for k, _ in model.named_parameters():
        d = (sd[k].detach().cpu().double() - o_sd[k].detach().double()).abs()
        assert d.max().item() <= 2.2 * lr, f"{what} {k}: max |d| {d.max().item():.3e} > {2.2 * lr:.1e}"
        rms = d.pow(2).mean().sqrt().item()
        assert rms <= 0.1 * lr, f"{what} {k}: RMS(d) {rms:.3e} > {0.1 * lr:.1e}"


def check_buffers_vs_oracle(model, o_sd, tol=1e-5, what=""):
    sd = model.state_dict()
    for k, v in sd.items():
        if "running" in k or "num_batches" in k:
            assert rel(v, o_sd[k]) < tol, f"{what} {k}"
