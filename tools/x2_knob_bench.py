"""A/B of one library tuning knob on the f32x2 GEMMs at the step's wide-layer shapes:
    python tools/x2_knob_bench.py <knob> <value> [<value> ...]      (value 0 = the library's default)"""
import os
os.environ.setdefault("USIP_ASSUME_LAUNCH_SAMPLES", "1")   # hand-built BatchNorm coefficients: the launch's own samples (usip_amd/ops.py::bound_covers)
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usip_amd import _lib, ops  # noqa: E402

dev = "cuda:0"
knob = sys.argv[1].encode()
values = [int(v) for v in sys.argv[2:]] or [0, 1]


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
for (M, K, P, nb) in [(512, 512, 8192, 16), (512, 256, 8192, 16), (256, 256, 8192, 16)]:
    At = (torch.randn(K, M, device=dev) * (2.0 / K) ** 0.5)
    X = torch.randn(nb, K, P, device=dev)
    b = torch.randn(M, device=dev)
    mu, var = X.mean(dim=(0, 2)), X.var(dim=(0, 2), unbiased=False)
    istd = torch.rsqrt(var + 1e-5)
    coef = torch.stack([istd, -mu * istd, mu, istd]).contiguous()
    G = torch.randn(nb, M, P, device=dev)
    Yg = torch.randn(nb, M, P, device=dev)
    mug, istdg = Yg.mean(dim=(0, 2)), torch.rsqrt(Yg.var(dim=(0, 2), unbiased=False) + 1e-5)
    cfw = torch.stack([istdg, -mug * istdg, mug, istdg]).contiguous()
    c4 = ops.bn_backward_reduce(G, Yg, cfw, mug, istdg, torch.ones(M, device=dev), True)[2]
    Wd = At.t().contiguous()
    ops.PLANES_CACHE = {}
    ref = refd = None
    for rnd in range(2):
        for v in values:
            _lib.lib().usip_set_tuning(knob, v)
            t1 = timed(lambda: ops.mlp_gemm(At, X, b, want_stats=True, pro=1, coef=coef))
            t2 = timed(lambda: ops.mlp_gemm(Wd, G, pro=2, X2=Yg, coef=c4, tag="dgrad")) if K <= 512 else 0.0
            y = ops.mlp_gemm(At, X, b, want_stats=True, pro=1, coef=coef)[0]
            d = ops.mlp_gemm(Wd, G, pro=2, X2=Yg, coef=c4, tag="dgrad")[0]
            if ref is None:
                ref, refd = y.clone(), d.clone()
            print("M=%d K=%d %s=%d: fwd+bnrelu %7.1f us  dgrad %7.1f us  bit-equal to first: %s %s  max diff / max: %.2e %.2e" % (
                M, K, sys.argv[1], v, t1, t2, bool(torch.equal(y, ref)), bool(torch.equal(d, refd)),
                float((y - ref).abs().max() / ref.abs().max()), float((d - refd).abs().max() / refd.abs().max())), flush=True)
    ops.PLANES_CACHE = None
_lib.lib().usip_set_tuning(knob, 0)
