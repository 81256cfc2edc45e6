"""f32x3 vs fp32-MFMA vs bf16 kernels at the detector's matrix-bound layer shapes (HIP events, median of 10), and
their error against an fp64 product on the same operands.

    python tools/x3_bench.py
"""
import os
os.environ.setdefault("USIP_ASSUME_LAUNCH_SAMPLES", "1")   # hand-built BatchNorm coefficients: the launch's own samples (usip_amd/ops.py::bound_covers)
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usip_amd import _lib, ops  # noqa: E402

dev = "cuda:0"


def timed(fn, it=10):
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
    return t[len(t) // 2] * 1e-3


def rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max())


print("device:", torch.cuda.get_device_name(0))
_lib.lib().usip_set_tuning(b"gemm_split3", 2)
for (M, K, P, nb) in [(512, 512, 8192, 16), (512, 256, 8192, 16), (256, 256, 8192, 16), (256, 131, 8192, 16),
                      (128, 128, 32768, 16), (128, 64, 32768, 16), (512, 640, 512, 16)]:
    At = (torch.randn(K, M, device=dev) * (2.0 / K) ** 0.5)
    X = torch.randn(nb, K, P, device=dev)
    b = torch.randn(M, device=dev)
    G = torch.randn(nb, M, P, device=dev)
    fl = 2.0 * M * K * P * nb
    truth = torch.matmul(At.double().t().unsqueeze(0), X[:1].double()) + b.double().view(1, M, 1)
    tw = torch.einsum("bmp,bnp->mn", G[:2].double(), X[:2].double())
    # pro = 1 operands as a training step has them: X is the pre-BatchNorm output of a layer, coef its (scale, shift,
    # mean, invstd) from the batch statistics (the f32x2 kernel derives its operand scale from them)
    mu, var = X.mean(dim=(0, 2)), X.var(dim=(0, 2), unbiased=False)
    istd = torch.rsqrt(var + 1e-5)
    gam, bet = 1 + 0.1 * torch.randn(K, device=dev), 0.1 * torch.randn(K, device=dev)
    coef = torch.stack([gam * istd, bet - mu * gam * istd, mu, istd]).contiguous()
    Yg = torch.randn(nb, M, P, device=dev)                      # the layer's own pre-BN output for the backward forms
    mug, varg = Yg.mean(dim=(0, 2)), Yg.var(dim=(0, 2), unbiased=False)
    istdg = torch.rsqrt(varg + 1e-5)
    cfw = torch.stack([istdg, -mug * istdg, mug, istdg]).contiguous()
    for mode in ("f32", "f32x3", "f32x2", "bf16"):
        ops.set_matmul_mode(mode)
        ops.PLANES_CACHE = {}                                   # weights are split once per step, not per launch
        t1 = timed(lambda: ops.mlp_gemm(At, X, b, want_stats=True))
        t2 = timed(lambda: ops.mlp_gemm(At, X, b, want_stats=True, pro=1, coef=coef))
        t3 = timed(lambda: ops.mlp_wgrad(G, X))
        c4 = ops.bn_backward_reduce(G, Yg, cfw, mug, istdg, torch.ones(M, device=dev), True)[2]
        Wd = At.t().contiguous()
        t4 = timed(lambda: ops.mlp_gemm(Wd, G, pro=2, X2=Yg, coef=c4, tag="dgrad")) if K <= 512 and M <= 512 else 0.0
        e1 = rel(ops.mlp_gemm(At, X[:1].contiguous(), b)[0], truth)
        act = torch.relu(torch.addcmul(coef[1].view(1, K, 1), X[:1], coef[0].view(1, K, 1)))    # fp32 prologue, as the kernels
        truth1 = torch.matmul(At.double().t().unsqueeze(0), act.double()) + b.double().view(1, M, 1)
        # (for the f32x2 scale the statistics must be those of the tensor the launch sees: use the full X)
        e2 = rel(ops.mlp_gemm(At, X, b, pro=1, coef=coef)[0][:1], truth1)
        e3 = rel(ops.mlp_wgrad(G[:2].contiguous(), X[:2].contiguous()), tw)
        print("M=%4d K=%4d P=%6d %-6s fwd %7.1f us %6.1f TF | fwd+bnrelu %7.1f us %6.1f TF | wgrad %7.1f us %6.1f TF | "
              "dgrad %7.1f us | err fwd %.1e fwd+bnrelu %.1e wgrad %.1e" % (M, K, P, mode, t1 * 1e6, fl / t1 / 1e12, t2 * 1e6,
                                                                    fl / t2 / 1e12, t3 * 1e6, fl / t3 / 1e12, t4 * 1e6, e1, e2, e3), flush=True)
    ops.set_matmul_mode("f32")
ops.PLANES_CACHE = None
_lib.lib().usip_set_tuning(b"gemm_split3", 0)
