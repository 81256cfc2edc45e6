"""The fused f32x2 layer backward (csrc/layer_bwd_x2.hip) against what it replaces, at the Ball detector's shapes
(16 clouds x 32768 positions): narrow_bwd.hip (fp32 MFMA) for the 64-input layers, the pooled data-gradient +
weight-gradient pair for conv5."""
import os
os.environ.setdefault("USIP_ASSUME_LAUNCH_SAMPLES", "1")   # hand-built BatchNorm coefficients: the launch's own samples (usip_amd/ops.py::bound_covers)
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usip_amd import _lib, ops  # noqa: E402

dev = "cuda:0"


def timed(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(it)]
    for s, e in evs:
        s.record()
        fn()
        e.record()
    torch.cuda.synchronize()
    t = sorted(s.elapsed_time(e) for s, e in evs)
    return t[len(t) // 2] * 1e3


def bn_inputs(nb, C, P):
    x = torch.randn(nb, C, P, device=dev)
    gamma, beta = 1 + 0.1 * torch.randn(C, device=dev), 0.1 * torch.randn(C, device=dev)
    mean, var = x.mean((0, 2)), x.var((0, 2), unbiased=False)
    invstd = torch.rsqrt(var + 1e-5)
    return x, gamma, mean.contiguous(), invstd.contiguous(), torch.stack([gamma * invstd, beta - mean * gamma * invstd, mean, invstd]).contiguous()


ops.set_matmul_mode("f32x2")
_lib.lib().usip_set_tuning(b"gemm_split3", 2)
nb, P = 16, 32768
ops.PLANES_CACHE = {}
for Cin, Cout in ((64, 64), (64, 128)):
    y, gy, my, iy, cy = bn_inputs(nb, Cout, P)
    x, gx, mx, ix, xcoef = bn_inputs(nb, Cin, P)
    dz = torch.randn(nb, Cout, P, device=dev)
    w2 = torch.randn(Cout, Cin, device=dev) * (2.0 / Cin) ** 0.5
    coef4 = ops.bn_backward_reduce(dz, y, cy, my, iy, gy, True)[2]
    t_old = timed(lambda: ops.mlp_narrow_backward(dz, y, coef4, x, xcoef, w2, want_red=True))
    if not ops.layer_backward_x2_supported(Cin, Cout, P, (dz, y, x), coef4, xcoef):
        print("%3d -> %3d: narrow_bwd (fp32 MFMA) %7.1f us | layer_bwd_x2 does not take this shape" % (Cin, Cout, t_old), flush=True)
        continue
    t_new = timed(lambda: ops.mlp_layer_backward_x2(dz, y, coef4, x, xcoef, w2, Cin=Cin, want_red=True))
    gb = 4.0 * nb * P * (2 * Cout + 2 * Cin) / 1e9
    print("%3d -> %3d: narrow_bwd (fp32 MFMA) %7.1f us %5.2f TB/s | layer_bwd_x2 %7.1f us %5.2f TB/s" % (
        Cin, Cout, t_old, gb / t_old * 1e3, t_new, gb / t_new * 1e3), flush=True)
# conv5: pooled form, 128 -> 128, K = 64 neighbours (the step's; 16 as the first argument for the other code path)
Cin = Cout = 128
K = int(sys.argv[1]) if len(sys.argv) > 1 else 64
M = P // K
y, gy, my, iy, cy = bn_inputs(nb, Cout, P)
x, gx, mx, ix, xcoef = bn_inputs(nb, Cin, P)
w2 = torch.randn(Cout, Cin, device=dev) * (2.0 / Cin) ** 0.5
pooled, arg = ops.group_max_act(y.view(nb, Cout, M, K), cy, True)
dpooled = torch.randn(nb, Cout, M, device=dev)
coef4 = ops.bn_pool_backward_reduce(dpooled, arg, y.view(nb, Cout, M, K), cy, my, iy, gy, True)[2]
pool = (dpooled, arg, K)
t_d = timed(lambda: ops.mlp_gemm(w2, None, pro=3, X2=y, coef=coef4, tag="dgrad", pool=pool))
t_w = timed(lambda: ops.mlp_wgrad(None, x, pro=3, G2=y, coef4=coef4, xcoef=xcoef, pool=pool))
t_new = timed(lambda: ops.mlp_layer_backward_x2(None, y, coef4, x, xcoef, w2, Cin=Cin, pool=pool))
if K % 32 == 0:
    t_red = timed(lambda: ops.mlp_layer_backward_x2(None, y, coef4, x, xcoef, w2, Cin=Cin, pool=pool, want_red=True, want_gsum=True))
    dx = ops.mlp_layer_backward_x2(None, y, coef4, x, xcoef, w2, Cin=Cin, pool=pool)[0]
    t_pass = timed(lambda: ops.bn_backward_reduce(dx, x, xcoef, mx, ix, gx, True, group=K))
    print("128 -> 128 pooled with the producing layer's sums: %7.1f us (the stand-alone pass it replaces: %7.1f us)" % (t_red, t_pass), flush=True)
gb = 4.0 * nb * P * (Cout + 2 * Cin) / 1e9
print("128 -> 128 pooled: dgrad %7.1f us + wgrad %7.1f us = %7.1f us | layer_bwd_x2 %7.1f us %5.2f TB/s" % (
    t_d, t_w, t_d + t_w, t_new, gb / t_new * 1e3), flush=True)
ops.PLANES_CACHE = None
_lib.lib().usip_set_tuning(b"gemm_split3", 0)
