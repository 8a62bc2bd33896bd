"""Are the PAIR launches of the BatchNorm backward (a downsample block's bn2 + the BatchNorm of its 1x1 branch in one reduction / one apply
pass, round 6) the same bit for bit as the two single launches they replace?  fp32 (float4 and rows kernels, packed bits and saved
activation) and bf16 (byte mask and saved activation).  usage (GPU box): python tools/probes/bn_pair_probe.py"""
import sys, torch
sys.path.insert(0, "semi-seg-ecg_amd")
from ssecg import ops
dev = torch.device("cuda:0")
for (N, C, L) in ((6, 128, 125), (6, 64, 500), (8, 256, 32), (4, 512, 63), (6, 128, 250)):
    g = torch.Generator(device=dev).manual_seed(N + C + L)
    x = torch.randn((N, C, L), generator=g, device=dev); x2 = torch.randn((N, C, L), generator=g, device=dev)
    dy = torch.randn((N, C, L), generator=g, device=dev); res = torch.randn((N, C, L), generator=g, device=dev)
    mean, invstd = 0.1 * torch.randn(C, generator=g, device=dev), 1 + 0.1 * torch.rand(C, generator=g, device=dev)
    mean2, invstd2 = 0.1 * torch.randn(C, generator=g, device=dev), 1 + 0.1 * torch.rand(C, generator=g, device=dev)
    gam, bet = 1 + 0.1 * torch.randn(C, generator=g, device=dev), 0.1 * torch.randn(C, generator=g, device=dev)
    gam2 = 1 + 0.1 * torch.randn(C, generator=g, device=dev)
    y, mask = ops.bn_apply_fwd(x, mean, invstd, gam, bet, res, True, want_mask=True)
    for name, ym in (("bits", mask), ("saved y", y)):
        pa = ops.bn_bwd_reduce(dy, ym, x, mean, invstd)
        pb = ops.bn_bwd_reduce(dy, ym, x2, mean2, invstd2)
        qa, qb = ops.bn_bwd_reduce_pair(dy, ym, x, mean, invstd, x2, mean2, invstd2)
        sa, sb = ops.bn_reduce_partials(pa), ops.bn_reduce_partials(pb)
        da, _ = ops.bn_bwd_apply(dy, ym, x, mean, invstd, gam, sa, N * L)
        db, _ = ops.bn_bwd_apply(dy, ym, x2, mean2, invstd2, gam2, sb, N * L)
        ea, eb = ops.bn_bwd_apply_pair(dy, ym, x, mean, invstd, gam, sa, x2, mean2, invstd2, gam2, sb, N * L)
        print((N, C, L), name, "reduce", torch.equal(pa, qa), torch.equal(pb, qb), float((pb - qb).abs().max()),
              "apply", torch.equal(da, ea), torch.equal(db, eb), float((da - ea).abs().max()), float((db - eb).abs().max()))

from ssecg import amp as SAMP
from ssecg import functional as SF
import types
for (N, C, L) in ((6, 128, 125), (8, 256, 32), (4, 512, 63)):
    g = torch.Generator(device=dev).manual_seed(7 * N + C + L)
    mk = lambda: SAMP.to_blocked(torch.randn((N, C, L), generator=g, device=dev))
    x, x2, dy, res = mk(), mk(), mk(), mk()
    mean, invstd = 0.1 * torch.randn(C, generator=g, device=dev), 1 + 0.1 * torch.rand(C, generator=g, device=dev)
    mean2, invstd2 = 0.1 * torch.randn(C, generator=g, device=dev), 1 + 0.1 * torch.rand(C, generator=g, device=dev)
    gam, bet = 1 + 0.1 * torch.randn(C, generator=g, device=dev), 0.1 * torch.randn(C, generator=g, device=dev)
    gam2 = 1 + 0.1 * torch.randn(C, generator=g, device=dev)
    y, mask = SAMP.bn_apply_fwd(x, mean, invstd, gam, bet, res, True, want_mask=True)
    for name, ym, mode in (("byte mask", mask, 3), ("saved y", y, 1)):
        pa = SAMP.bn_bwd_reduce(dy, ym, x, mean, invstd, gam, bet, mode)
        pb = SAMP.bn_bwd_reduce(dy, ym, x2, mean2, invstd2, gam2, bet, mode)
        sa, sb = ops.bn_reduce_partials(pa), ops.bn_reduce_partials(pb)
        da, _ = SAMP.bn_bwd_apply(dy, ym, x, mean, invstd, gam, bet, mode, sa, N * L)
        db, _ = SAMP.bn_bwd_apply(dy, ym, x2, mean2, invstd2, gam2, bet, mode, sb, N * L)
        u2 = types.SimpleNamespace(c=x, y=ym, mean=mean, invstd=invstd, gamma=gam, group=None, count=N * L)
        ud = types.SimpleNamespace(c=x2, mean=mean2, invstd=invstd2, gamma=gam2, group=None, count=N * L)
        ea, eb, dg2, db2, dgd, dbd = SAMP.bn_bwd_pair(u2, ud, dy)
        _, dga, dba = ops.bn_reduce_partials(pa, want_param_grads=True)
        _, dgb, dbb = ops.bn_reduce_partials(pb, want_param_grads=True)
        print("bf16", (N, C, L), name, "apply", torch.equal(da, ea), torch.equal(db, eb), "param grads", torch.equal(dga, dg2), torch.equal(dgb, dgd),
              torch.equal(dba, db2), torch.equal(dbb, dbd))
