#!/usr/bin/env python3
"""How often does a bf16 rounding land on the other side between two correct evaluations of the same convolution?
HIP (v_mfma_f32_32x32x16_bf16, fp32 accumulation inside the matrix pipe) vs torch-CPU fp32 vs torch-CPU fp64 accumulation, same
bf16 operands.  Prints the fraction of outputs that differ after rounding to bf16 and the pre-rounding relative deviation -
the implementation-noise floor the model-level bf16 tests are calibrated against (tests/test_amp_gpu.py)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "semi-seg-ecg_amd"), os.path.join(ROOT, "tests")]
import torch
import torch.nn.functional as F

from ssecg import amp as SAMP
from ssecg import ops, synth

dev = torch.device("cuda:0")
rb = lambda t: t.to(torch.bfloat16).to(torch.float32)
rnd = lambda seed, *shape, std=1.0: torch.from_numpy(synth.normal(seed, 9, shape, std=std))
for (N, Cin, Lin, Cout, K, s, p) in [(4, 64, 500, 64, 3, 1, 1), (4, 64, 500, 128, 3, 2, 1), (4, 128, 250, 128, 3, 1, 1),
                                    (4, 256, 125, 256, 3, 1, 1), (4, 512, 63, 512, 3, 1, 1)]:
    x = rb(torch.relu(rnd(1, N, Cin, Lin)))
    w = rnd(2, Cout, Cin, K, std=(2.0 / (K * Cout)) ** 0.5)
    y32 = F.conv1d(x, rb(w), stride=s, padding=p)
    y64 = F.conv1d(x.double(), rb(w).double(), stride=s, padding=p)
    ops.begin_forward()
    yb, _ = SAMP.conv_fwd(SAMP.to_blocked(x.to(dev)), w.to(dev), s, p, want_stats=True)
    yh = SAMP.to_planar(yb).cpu()
    r32, r64 = rb(y32), rb(y64.float())
    print(f"conv {Cin}->{Cout} k{K} s{s} L{Lin}: bf16 outputs differing  HIP vs cpu-fp64 {(yh != r64).float().mean():.3%}  "
          f"cpu-fp32 vs cpu-fp64 {(r32 != r64).float().mean():.3%};  pre-rounding cpu-fp32 vs fp64 rel. L2 "
          f"{((y32.double() - y64).norm() / y64.norm()).item():.1e}")

# ---- BatchNorm (+residual, ReLU) forward and backward on blocked bf16: the same question for the element-wise kernels
for (N, C, L, relu, use_res) in [(4, 64, 500, True, False), (4, 128, 250, True, True), (4, 128, 250, False, False), (4, 512, 63, True, True)]:
    x = rb(rnd(1, N, C, L) * 1.7 + 0.4).requires_grad_(True)
    g = (1.0 + 0.2 * rnd(2, C)).requires_grad_(True); b = (0.1 * rnd(3, C)).requires_grad_(True)
    res = rb(rnd(4, N, C, L)).requires_grad_(True) if use_res else None
    z = F.batch_norm(x, None, None, g, b, training=True, momentum=0.1, eps=1e-5)
    if use_res: z = z + res
    y_ref = F.relu(z) if relu else z
    dy = rb(rnd(7, N, C, L))
    grads = torch.autograd.grad(y_ref, (x, g, b) + ((res,) if use_res else ()), dy)
    xd = x.detach().double()
    mean = xd.mean(dim=(0, 2)).float().to(dev)
    invstd = (xd.var(dim=(0, 2), unbiased=False) + 1e-5).rsqrt().float().to(dev)
    xb = SAMP.to_blocked(x.detach().to(dev))
    gg, bg = g.detach().to(dev), b.detach().to(dev)
    resb = SAMP.to_blocked(res.detach().to(dev)) if use_res else None
    yb = SAMP.bn_apply_fwd(xb, mean, invstd, gg, bg, resb, relu)
    yh = SAMP.to_planar(yb).cpu()
    dyb = SAMP.to_blocked(dy.to(dev))
    mode = 0 if not relu else (1 if use_res else 2)
    part = SAMP.bn_bwd_reduce(dyb, yb if mode == 1 else None, xb, mean, invstd, gg, bg, mode)
    sums, dgam, dbet = ops.bn_reduce_partials(part, want_param_grads=True)
    dx, dz = SAMP.bn_bwd_apply(dyb, yb if mode == 1 else None, xb, mean, invstd, gg, bg, mode, sums, N * L, want_dz=use_res)
    dxh = SAMP.to_planar(dx).cpu()
    ref_dx = grads[0]
    print(f"bn C={C} L={L} relu={relu} res={use_res}: forward outputs differing {(yh != rb(y_ref.detach())).float().mean():.3%}; "
          f"backward dx differing from rb(torch) {(dxh != rb(ref_dx)).float().mean():.3%}, rel. L2 vs rb(torch) "
          f"{((dxh - rb(ref_dx)).norm() / ref_dx.norm()).item():.2e}, vs unrounded {((dxh - ref_dx).norm() / ref_dx.norm()).item():.2e}; "
          f"dgamma rel {((dgam.cpu() - grads[1]).abs().max() / grads[1].abs().max()).item():.1e}")
