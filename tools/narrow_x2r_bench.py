"""The narrow forward layers (64 -> 64, 64 -> 128; BN+ReLU prologue, statistics) on the streaming f32x3 kernel
(narrow_fwd.hip) against the register-resident f32x2 kernel (gemm_x2r_kernel), same box, same operands."""
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


ops.set_matmul_mode("f32x2")
_lib.lib().usip_set_tuning(b"gemm_split3", 2)
for (M, K, P, nb) in [(64, 64, 524288, 16), (128, 64, 524288, 16), (128, 128, 524288, 16)]:
    At = (torch.randn(K, M, device=dev) * (2.0 / K) ** 0.5)
    X = torch.randn(nb, K, P, device=dev)
    b = torch.randn(M, device=dev)
    mu, var = X.mean(dim=(0, 2)), X.var(dim=(0, 2), unbiased=False)
    istd = torch.rsqrt(var + 1e-5)
    gam, bet = 1 + 0.1 * torch.randn(K, device=dev), 0.1 * torch.randn(K, device=dev)
    coef = torch.stack([gam * istd, bet - mu * gam * istd, mu, istd]).contiguous()
    act = torch.relu(torch.addcmul(coef[1].view(1, K, 1), X[:1, :, :4096], coef[0].view(1, K, 1)))
    truth = torch.matmul(At.double().t().unsqueeze(0), act.double()) + b.double().view(1, M, 1)
    ops.PLANES_CACHE = {}
    for narrow in (True, False):
        ops.NARROW_FWD = narrow
        t = timed(lambda: ops.mlp_gemm(At, X, b, want_stats=True, pro=1, coef=coef))
        y, st = ops.mlp_gemm(At, X, b, want_stats=True, pro=1, coef=coef)
        err = float((y[:1, :, :4096].double() - truth).abs().max() / truth.abs().max())
        s1 = st[0].double().sum(dim=1)
        es = float((s1 - y.double().sum(dim=(0, 2))).abs().max() / y.double().abs().sum(dim=(0, 2)).max())
        gb = 4.0 * nb * P * (K + M) / 1e9
        print("M=%3d K=%3d positions=%d narrow_fwd=%s: %7.1f us  %5.2f TB/s  err %.1e  stats err %.1e" % (
            M, K, nb * P, narrow, t, gb / t * 1e-3 * 1e3, err, es), flush=True)
    ops.PLANES_CACHE = None
ops.NARROW_FWD = True
_lib.lib().usip_set_tuning(b"gemm_split3", 0)
