"""A few launches of the register-resident f32x2 GEMM (128 x 128 over 524288 positions, BN+ReLU prologue, statistics;
then its data-gradient form) for counter passes."""
import os
os.environ.setdefault("USIP_ASSUME_LAUNCH_SAMPLES", "1")   # hand-built BatchNorm coefficients: the launch's own samples (usip_amd/ops.py::bound_covers)
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usip_amd import _lib, ops  # noqa: E402

dev = "cuda:0"
ops.set_matmul_mode("f32x2")
_lib.lib().usip_set_tuning(b"gemm_split3", 2)
M, K, P, nb = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (128, 128, 32768, 16)
At = (torch.randn(K, M, device=dev) * (2.0 / K) ** 0.5)
X = torch.randn(nb, K, P, device=dev)
b = torch.randn(M, device=dev)
mu, var = X.mean(dim=(0, 2)), X.var(dim=(0, 2), unbiased=False)
istd = torch.rsqrt(var + 1e-5)
coef = torch.stack([istd, -mu * istd, mu, istd]).contiguous()
ops.PLANES_CACHE = {}
ops.NARROW_FWD = False
for _ in range(6):
    ops.mlp_gemm(At, X, b, want_stats=True, pro=1, coef=coef)
torch.cuda.synchronize()
