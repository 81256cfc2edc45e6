import os
os.environ.setdefault("USIP_ASSUME_LAUNCH_SAMPLES", "1")   # hand-built BatchNorm coefficients: the launch's own samples (usip_amd/ops.py::bound_covers), sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usip_amd import _lib, ops  # noqa: E402
dev = "cuda:0"
ops.set_matmul_mode("f32x2")
_lib.lib().usip_set_tuning(b"gemm_split3", 2)
for (nb, K, M, P) in [(4, 256, 256, 8192), (4, 272, 256, 8192), (4, 32, 256, 8192), (4, 64, 256, 8192), (16, 512, 512, 8192)]:
    torch.manual_seed(0)
    At = (torch.randn(K, M, device=dev) * (2.0 / K) ** 0.5)
    X = torch.randn(nb, K, P, device=dev)
    mu, var = X.mean(dim=(0, 2)), X.var(dim=(0, 2), unbiased=False)
    istd = torch.rsqrt(var + 1e-5)
    coef = torch.stack([istd, -mu * istd, mu, istd]).contiguous()
    for knob in (0, 2):
        for st in (False, True):
            _lib.lib().usip_set_tuning(b"x2_direct", 1)
            ref = ops.mlp_gemm(At, X, None, want_stats=st, pro=1, coef=coef)
            _lib.lib().usip_set_tuning(b"x2_direct", knob)
            y = ops.mlp_gemm(At, X, None, want_stats=st, pro=1, coef=coef)
            d = (y[0] - ref[0]).abs() / ref[0].abs().max()
            print("K=%d nb=%d knob %d stats %s: max diff %.2e bad frac %.4f" % (K, nb, knob, st, float(d.max()), float((d > 1e-5).float().mean())))
