import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usip_amd import ops
dev = "cuda:0"
def bn_inputs(nb, C, P):
    x = torch.randn(nb, C, P, device=dev)
    gamma, beta = 1 + 0.1 * torch.randn(C, device=dev), 0.1 * torch.randn(C, device=dev)
    mean, var = x.mean((0, 2)), x.var((0, 2), unbiased=False)
    invstd = torch.rsqrt(var + 1e-5)
    return x, gamma, mean.contiguous(), invstd.contiguous(), torch.stack([gamma * invstd, beta - mean * gamma * invstd, mean, invstd]).contiguous()
ops.set_matmul_mode("f32x2")
torch.manual_seed(3)
Cin, Cout, nb, P, Ctot, wcol = 64, 128, 4, 8192, 128, 64
y, gy, my, iy, cy = bn_inputs(nb, Cout, P)
x, gx, mx, ix, xcoef = bn_inputs(nb, Cin, P)
dz = torch.randn(nb, Cout, P, device=dev)
w2 = torch.randn(Cout, Ctot, device=dev) * (2.0 / Cin) ** 0.5
coef4 = ops.bn_backward_reduce(dz, y, cy, my, iy, gy, True)[2]
c = [coef4[i].double().view(1, Cout, 1) for i in range(4)]
fma = lambda a_, b_, c_: (a_.double() * b_.double() + c_.double()).float()
dyh = torch.where(fma(y, c[0], c[1]) > 0, dz, torch.zeros_like(dz))
dy = fma(c[0], dyh, fma(c[2], y, c[3])).double()
want_dx = torch.einsum("oc,bop->bcp", w2[:, wcol:wcol + Cin].double(), dy)
for cached in (False, True):
    ops.PLANES_CACHE = {} if cached else None
    outs = []
    for i in range(8):
        res = ops.mlp_layer_backward_x2(dz, y, coef4, x, xcoef, w2, wcol=wcol, dw_out=torch.zeros(Cout, Ctot, device=dev), Cin=Cin, want_red=False)
        outs.append(res[0].clone())
    tpc = P // 32
    print("planes cached:", cached)
    for i, o in enumerate(outs):
        err = (o.double() - want_dx).abs()
        bad = (err > 2e-5).view(nb, Cin, tpc, 32)
        tiles = bad.any(3).any(1)
        idx = tiles.nonzero()
        T = idx[:, 0] * tpc + idx[:, 1]
        rows = bad.any(3).any(2).any(0).nonzero().flatten().tolist()
        poss = bad.any(2).any(1).any(0).nonzero().flatten().tolist()
        print(" launch %d: max err %.3e, elements off %d, bad tiles %d of %d; t: %s; WGs: %d distinct (first %s); ci rows %d (%s..%s); pos-in-tile %d" % (
            i, float(err.max()), int(bad.sum()), int(tiles.sum()), nb * tpc, sorted(set((T // 512).tolist())), len(set((T % 512).tolist())),
            sorted(set((T % 512).tolist()))[:8], len(rows), rows[:3], rows[-3:], len(poss)))
ops.PLANES_CACHE = None
