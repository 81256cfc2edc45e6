"""The narrow forward GEMMs of the ball front end alone (64 -> 64 and 64 -> 128 at 524288 positions), for counters."""
import os
import sys
os.environ.setdefault("USIP_ASSUME_LAUNCH_SAMPLES", "1")   # hand-built BatchNorm coefficients: the launch's own samples (usip_amd/ops.py::bound_covers), sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usip_amd import ops
dev = "cuda:0"
nb, P = 16, 32768
for (M, K) in [(64, 64), (128, 64)]:
    At = torch.randn(K, M, device=dev) * 0.1
    b = torch.randn(M, device=dev)
    coef = torch.stack([1 + 0.1 * torch.randn(K, device=dev), 0.1 * torch.randn(K, device=dev)]).contiguous()
    ring = [torch.randn(nb, K, P, device=dev) for _ in range(3)]
    for i in range(6):
        ops.mlp_gemm(At, ring[i % 3], b, want_stats=True, pro=1, coef=coef)
torch.cuda.synchronize()
