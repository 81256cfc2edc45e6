"""The fused f32x2 layer backward launched N times on the same inputs: every output must be bit-identical.
  python tools/layer_bwd_race.py [N]"""
import os
os.environ.setdefault("USIP_ASSUME_LAUNCH_SAMPLES", "1")   # hand-built BatchNorm coefficients: the launch's own samples (usip_amd/ops.py::bound_covers)
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usip_amd import ops  # noqa: E402

dev = "cuda:0"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200


def bn_inputs(nb, C, P):
    x = torch.randn(nb, C, P, device=dev)
    gamma, beta = 1 + 0.1 * torch.randn(C, device=dev), 0.1 * torch.randn(C, device=dev)
    mean, var = x.mean((0, 2)), x.var((0, 2), unbiased=False)
    invstd = torch.rsqrt(var + 1e-5)
    return x, gamma, mean.contiguous(), invstd.contiguous(), torch.stack([gamma * invstd, beta - mean * gamma * invstd, mean, invstd]).contiguous()


ops.set_matmul_mode("f32x2")
torch.manual_seed(3)
for Cin, Cout, nb, P, Ctot, wcol in ((64, 64, 4, 4096, 64, 0), (64, 64, 16, 32768, 128, 64), (64, 128, 4, 4096, 128, 64), (64, 128, 16, 32768, 128, 64),
                                     (64, 128, 8, 16384, 128, 0)):
    y, gy, my, iy, cy = bn_inputs(nb, Cout, P)
    x, gx, mx, ix, xcoef = bn_inputs(nb, Cin, P)
    dz = torch.randn(nb, Cout, P, device=dev)
    w2 = torch.randn(Cout, Ctot, device=dev) * (2.0 / Cin) ** 0.5
    coef4 = ops.bn_backward_reduce(dz, y, cy, my, iy, gy, True)[2]
    for red in (True, False):
        ref = None
        bad = 0
        for i in range(N):
            dw_out = torch.zeros(Cout, Ctot, device=dev)
            res = ops.mlp_layer_backward_x2(dz, y, coef4, x, xcoef, w2, wcol=wcol, dw_out=dw_out, Cin=Cin, want_red=red)
            cur = [res[0], res[1]] + ([res[2].flat] if red else [])
            if ref is None:
                ref = [t.clone() for t in cur]
            elif not all(torch.equal(a, b) for a, b in zip(ref, cur)):
                bad += 1
                if bad <= 2:
                    print("   differs:", [("dx", "dw", "red")[k] for k, (a, b) in enumerate(zip(ref, cur)) if not torch.equal(a, b)],
                          [float((a - b).abs().max()) for a, b in zip(ref, cur)])
        print("%3d -> %3d nb %2d P %5d red %-5s: %d of %d launches differ" % (Cin, Cout, nb, P, red, bad, N), flush=True)

# pooled form (128 -> 128): one 8-wave workgroup per CU; with and without the producing layer's sums
for nb, M, K in ((16, 512, 64), (4, 256, 32), (16, 2048, 16)):
    Cin = Cout = 128
    P = M * K
    y, gy, my, iy, cy = bn_inputs(nb, Cout, P)
    x, gx, mx, ix, xcoef = bn_inputs(nb, Cin, P)
    w2 = torch.randn(Cout, Cin, device=dev) * (2.0 / Cin) ** 0.5
    pooled, arg = ops.group_max_act(y.view(nb, Cout, M, K), cy, True)
    dpooled = torch.randn(nb, Cout, M, device=dev)
    coef4 = ops.bn_pool_backward_reduce(dpooled, arg, y.view(nb, Cout, M, K), cy, my, iy, gy, True)[2]
    for red in ((True, False) if K % 32 == 0 else (False,)):
        ref = None
        bad = 0
        for i in range(N):
            res = ops.mlp_layer_backward_x2(None, y, coef4, x, xcoef, w2, Cin=Cin, pool=(dpooled, arg, K), want_red=red, want_gsum=red)
            cur = [res[0], res[1]] + ([res[2].flat, res[2].gsum] if red else [])
            if ref is None:
                ref = [t.clone() for t in cur]
            elif not all(torch.equal(a, b) for a, b in zip(ref, cur)):
                bad += 1
        print("128 -> 128 pooled nb %2d M %4d K %2d red %-5s: %d of %d launches differ" % (nb, M, K, red, bad, N), flush=True)
