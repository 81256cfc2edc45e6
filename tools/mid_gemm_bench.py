"""The second-stage / head layers (512 positions per cloud, 16 clouds): fp32 MFMA kernel against the f32x2 tile kernel
with 256-row and 128-row tiles.  python tools/mid_gemm_bench.py"""
import os
os.environ.setdefault("USIP_ASSUME_LAUNCH_SAMPLES", "1")   # hand-built BatchNorm coefficients: the launch's own samples (usip_amd/ops.py::bound_covers)
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usip_amd import _lib, ops  # noqa: E402

dev = "cuda:0"


def timed(fn, it=30):
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


lib = _lib.lib()
for (M, K, P, nb) in [(512, 640, 512, 16), (512, 512, 512, 16), (256, 512, 512, 16), (640, 512, 512, 16), (512, 256, 512, 16),
                      (256, 256, 8192, 16)]:
    At = (torch.randn(K, M, device=dev) * (2.0 / K) ** 0.5)
    X = torch.randn(nb, K, P, device=dev)
    b = torch.randn(M, device=dev)
    mu, var = X.mean(dim=(0, 2)), X.var(dim=(0, 2), unbiased=False)
    istd = torch.rsqrt(var + 1e-5)
    coef = torch.stack([istd, -mu * istd, mu, istd]).contiguous()
    out = []
    for mode, split, tile in (("f32", 0, 0), ("f32x2", 2, 0), ("f32x2", 2, 1), ("f32x3", 2, 0)):
        ops.set_matmul_mode(mode)
        lib.usip_set_tuning(b"gemm_split3", split)
        lib.usip_set_tuning(b"x3_gemm_tile", tile)
        ops.PLANES_CACHE = {}
        t = timed(lambda: ops.mlp_gemm(At, X, b, want_stats=True, pro=1, coef=coef)) if K <= 640 else 0.0
        out.append("%s%s %6.1f us" % (mode, " 128-row" if tile else "", t))
        ops.PLANES_CACHE = None
    print("M=%d K=%d P=%d x %d fwd+bnrelu: %s" % (M, K, P, nb, " | ".join(out)), flush=True)
lib.usip_set_tuning(b"gemm_split3", 0)
lib.usip_set_tuning(b"x3_gemm_tile", 0)
ops.set_matmul_mode("f32")

print("weight gradients (pro = 2: dY rebuilt from (dZ, Y); X through its BN+ReLU prologue)")
for (M, N, P, nb) in [(512, 640, 512, 16), (512, 512, 512, 16), (256, 512, 512, 16), (512, 256, 512, 16), (256, 256, 512, 16),
                      (128, 128, 512, 16)]:
    G = torch.randn(nb, M, P, device=dev)
    Yg = torch.randn(nb, M, P, device=dev)
    X = torch.randn(nb, N, P, device=dev)
    mug, istdg = Yg.mean(dim=(0, 2)), torch.rsqrt(Yg.var(dim=(0, 2), unbiased=False) + 1e-5)
    cfw = torch.stack([istdg, -mug * istdg, mug, istdg]).contiguous()
    mux, istdx = X.mean(dim=(0, 2)), torch.rsqrt(X.var(dim=(0, 2), unbiased=False) + 1e-5)
    xcoef = torch.stack([istdx, -mux * istdx, mux, istdx]).contiguous()
    out = []
    for mode, split in (("f32", 0), ("f32x2", 0), ("f32x2", 2), ("f32x3", 2)):
        ops.set_matmul_mode(mode)
        lib.usip_set_tuning(b"gemm_split3", split)
        c4 = ops.bn_backward_reduce(G, Yg, cfw, mug, istdg, torch.ones(M, device=dev), True)[2]
        t = timed(lambda: ops.mlp_wgrad(G, X, pro=2, G2=Yg, coef4=c4, xcoef=xcoef))
        out.append("%s%s %6.1f us" % (mode, " forced" if split else "", t))
    print("M=%d N=%d P=%d x %d wgrad: %s" % (M, N, P, nb, " | ".join(out)), flush=True)
lib.usip_set_tuning(b"gemm_split3", 0)
ops.set_matmul_mode("f32")
