"""A few launches of one form of the fused f32x2 layer backward for counter passes: layer_bwd_one.py <Cin> <Cout> [pool]"""
import os
os.environ.setdefault("USIP_ASSUME_LAUNCH_SAMPLES", "1")   # hand-built BatchNorm coefficients: the launch's own samples (usip_amd/ops.py::bound_covers)
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usip_amd import _lib, ops  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
dev = "cuda:0"
Cin, Cout = int(sys.argv[1]), int(sys.argv[2])
pooled = len(sys.argv) > 3
nb, P, K = 16, 32768, 64


def bn_inputs(nb, C, P):
    x = torch.randn(nb, C, P, device=dev)
    gamma, beta = 1 + 0.1 * torch.randn(C, device=dev), 0.1 * torch.randn(C, device=dev)
    mean, var = x.mean((0, 2)), x.var((0, 2), unbiased=False)
    invstd = torch.rsqrt(var + 1e-5)
    return x, gamma, mean.contiguous(), invstd.contiguous(), torch.stack([gamma * invstd, beta - mean * gamma * invstd, mean, invstd]).contiguous()


ops.set_matmul_mode("f32x2")
ops.PLANES_CACHE = {}
y, gy, my, iy, cy = bn_inputs(nb, Cout, P)
x, gx, mx, ix, xcoef = bn_inputs(nb, Cin, P)
w2 = torch.randn(Cout, Cin, device=dev) * (2.0 / Cin) ** 0.5
if pooled:
    M = P // K
    _, arg = ops.group_max_act(y.view(nb, Cout, M, K), cy, True)
    dpooled = torch.randn(nb, Cout, M, device=dev)
    coef4 = ops.bn_pool_backward_reduce(dpooled, arg, y.view(nb, Cout, M, K), cy, my, iy, gy, True)[2]
    for _ in range(6):
        ops.mlp_layer_backward_x2(None, y, coef4, x, xcoef, w2, Cin=Cin, pool=(dpooled, arg, K))
else:
    dz = torch.randn(nb, Cout, P, device=dev)
    coef4 = ops.bn_backward_reduce(dz, y, cy, my, iy, gy, True)[2]
    for _ in range(6):
        ops.mlp_layer_backward_x2(dz, y, coef4, x, xcoef, w2, Cin=Cin, want_red=True)
torch.cuda.synchronize()
