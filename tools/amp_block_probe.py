#!/usr/bin/env python3
"""Where do the HIP bf16 path and oracle/amp_ref.py part ways inside ONE stride-2 BasicBlock?  Every intermediate tensor of
the forward, HIP pieces (ssecg.amp) vs emulation pieces, fed the SAME inputs at every stage (so each line shows that op alone)
and chained (so the last lines show the accumulated effect)."""
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
blk = lambda t: SAMP.to_blocked(t.to(dev))
pl = lambda t: SAMP.to_planar(t).cpu()


def cmp(name, hip, ref):
    d = (hip != ref).float().mean().item()
    print(f"  {name:34s} differing {d:.4%}   rel. L2 {((hip - ref).norm() / ref.norm()).item():.2e}")


def bn_train(c, g, b):
    m = c.double().mean(dim=(0, 2)); v = c.double().var(dim=(0, 2), unbiased=False)
    return F.batch_norm(c, None, None, g, b, training=True, eps=1e-5), m.float(), (v + 1e-5).rsqrt().float()


for (cin, cout, stride, L, N) in [(64, 128, 2, 500, 3), (64, 64, 1, 500, 3)]:
    print(f"block {cin}->{cout} stride {stride} L {L}")
    x = rb(torch.relu(rnd(1, N, cin, L)))
    w1 = rnd(2, cout, cin, 3, std=(2 / (3 * cout)) ** 0.5); w2 = rnd(3, cout, cout, 3, std=(2 / (3 * cout)) ** 0.5)
    wd = rnd(4, cout, cin, 1, std=(2 / cout) ** 0.5)
    g1, b1 = 1 + 0.2 * rnd(5, cout), 0.1 * rnd(6, cout)
    g2, b2 = 1 + 0.2 * rnd(7, cout), 0.1 * rnd(8, cout)
    gd, bd = 1 + 0.2 * rnd(9, cout), 0.1 * rnd(10, cout)
    ops.begin_forward()
    # conv1
    c1_ref = rb(F.conv1d(x, rb(w1), stride=stride, padding=1))
    c1b, part = SAMP.conv_fwd(blk(x), w1.to(dev), stride, 1, want_stats=True)
    cmp("conv1 output", pl(c1b), c1_ref)
    # bn1 statistics from the HIP partial sums vs fp64 statistics of the reference tensor
    z_ref, m_ref, i_ref = bn_train(c1_ref, g1, b1)
    dummy_rm, dummy_rv = torch.zeros(cout, device=dev), torch.ones(cout, device=dev)
    mean, invstd = ops.bn_stats_finalize(part, c1_ref.shape[0] * c1_ref.shape[2], 1e-5, 0.1, dummy_rm, dummy_rv)
    print(f"  bn1 mean max|d| {(mean.cpu() - m_ref).abs().max().item():.2e} (|mean| max {m_ref.abs().max().item():.2e}), invstd rel "
          f"{((invstd.cpu() - i_ref).abs() / i_ref).max().item():.2e}")
    a1_ref = rb(F.relu(z_ref))
    a1b = SAMP.bn_apply_fwd(blk(c1_ref), mean, invstd, g1.to(dev), b1.to(dev), None, True)
    cmp("bn1+relu (HIP stats, same c1)", pl(a1b), a1_ref)
    a1b2 = SAMP.bn_apply_fwd(blk(c1_ref), m_ref.to(dev), i_ref.to(dev), g1.to(dev), b1.to(dev), None, True)
    cmp("bn1+relu (fp64 stats, same c1)", pl(a1b2), a1_ref)
    # downsample branch
    if stride != 1 or cin != cout:
        cd_ref = rb(F.conv1d(x, rb(wd), stride=stride))
        cdb, partd = SAMP.conv_fwd(blk(x), wd.to(dev), stride, 0, want_stats=True)
        cmp("downsample conv output", pl(cdb), cd_ref)
        zd_ref, md_ref, id_ref = bn_train(cd_ref, gd, bd)
        md, idd = ops.bn_stats_finalize(partd, cd_ref.shape[0] * cd_ref.shape[2], 1e-5, 0.1, dummy_rm, dummy_rv)
        print(f"  bn_ds mean max|d| {(md.cpu() - md_ref).abs().max().item():.2e} (|mean| max {md_ref.abs().max().item():.2e}), invstd rel "
              f"{((idd.cpu() - id_ref).abs() / id_ref).max().item():.2e}")
        idt_ref = rb(zd_ref)
        idtb = SAMP.bn_apply_fwd(blk(cd_ref), md, idd, gd.to(dev), bd.to(dev), None, False)
        cmp("bn_ds (HIP stats, same c)", pl(idtb), idt_ref)
    else:
        idt_ref = x
    c2_ref = rb(F.conv1d(a1_ref, rb(w2), padding=1))
    c2b, part2 = SAMP.conv_fwd(blk(a1_ref), w2.to(dev), 1, 1, want_stats=True)
    cmp("conv2 output", pl(c2b), c2_ref)
    z2_ref, m2_ref, i2_ref = bn_train(c2_ref, g2, b2)
    m2, i2 = ops.bn_stats_finalize(part2, c2_ref.shape[0] * c2_ref.shape[2], 1e-5, 0.1, dummy_rm, dummy_rv)
    out_ref = rb(F.relu(z2_ref + idt_ref))
    outb = SAMP.bn_apply_fwd(blk(c2_ref), m2, i2, g2.to(dev), b2.to(dev), blk(idt_ref), True)
    cmp("bn2 + residual + relu", pl(outb), out_ref)
