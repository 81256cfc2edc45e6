"""Stand-alone timing of the shared-MLP GEMM / wgrad kernels at the detector's layer shapes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usip_amd import ops


def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e-3


dev = "cuda:0"
for (M, K, P, nb) in [(512, 512, 8192, 16), (256, 256, 8192, 16), (128, 128, 32768, 16), (64, 64, 32768, 16), (512, 640, 512, 16)]:
    At = torch.randn(K, M, device=dev); X = torch.randn(nb, K, P, device=dev); b = torch.randn(M, device=dev)
    dt = t(lambda: ops.mlp_gemm(At, X, b, want_stats=True))
    fl = 2.0 * M * K * P * nb
    print("gemm fwd  M=%4d K=%4d P=%6d: %8.1f us  %6.1f TFLOP/s" % (M, K, P, dt * 1e6, fl / dt / 1e12))
    G = torch.randn(nb, M, P, device=dev)
    dt = t(lambda: ops.mlp_wgrad(G, X))
    print("wgrad     M=%4d N=%4d P=%6d: %8.1f us  %6.1f TFLOP/s" % (M, K, P, dt * 1e6, fl / dt / 1e12))
