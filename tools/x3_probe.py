"""One kernel under the profiler: the 512x512 forward GEMM (BN+ReLU prologue, statistics epilogue) in a chosen mode.
    python tools/x3_probe.py f32x3|f32|bf16 [M K P nb]"""
import os
os.environ.setdefault("USIP_ASSUME_LAUNCH_SAMPLES", "1")   # hand-built BatchNorm coefficients: the launch's own samples (usip_amd/ops.py::bound_covers)
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usip_amd import _lib, ops  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "f32x3"
M, K, P, nb = (int(v) for v in sys.argv[2:6]) if len(sys.argv) >= 6 else (512, 512, 8192, 16)
dev = "cuda:0"
ops.set_matmul_mode(mode)
_lib.lib().usip_set_tuning(b"gemm_split3", 2)
At = torch.randn(K, M, device=dev) * (2.0 / K) ** 0.5
X = torch.randn(nb, K, P, device=dev)
b = torch.randn(M, device=dev)
coef = torch.stack([1 + 0.1 * torch.randn(K, device=dev), 0.1 * torch.randn(K, device=dev)])
G = torch.randn(nb, M, P, device=dev)
for _ in range(5):
    ops.mlp_gemm(At, X, b, want_stats=True, pro=1, coef=coef)
    ops.mlp_wgrad(G, X)
torch.cuda.synchronize()
