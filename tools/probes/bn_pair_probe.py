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
