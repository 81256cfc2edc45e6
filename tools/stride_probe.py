"""Does the power-of-two row stride of the activation tensors (P = 32768 positions per cloud -> 128 KiB between the
channel rows of a tile) cost HBM channel conflicts?  Time the narrow forward GEMM and the BN-backward reduction at
P = 32768 and at slightly different P (same bytes to 1 %)."""
import os
os.environ.setdefault("USIP_ASSUME_LAUNCH_SAMPLES", "1")   # hand-built BatchNorm coefficients: the launch's own samples (usip_amd/ops.py::bound_covers), sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usip_amd import ops
dev = "cuda:0"
def timed(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(it)]
    for s, e in evs:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    t = sorted(s.elapsed_time(e) for s, e in evs)
    return t[len(t) // 2] * 1e3
nb = 16
for (M, K) in [(64, 64), (128, 64), (128, 128)]:
    for P in [32768, 32768 + 64, 32768 + 128, 32768 + 256, 32768 + 1024, 32768 - 256, 30000, 36864]:
        At = torch.randn(K, M, device=dev) * 0.1
        b = torch.randn(M, device=dev)
        coef = torch.stack([1 + 0.1 * torch.randn(K, device=dev), 0.1 * torch.randn(K, device=dev)]).contiguous()
        ring = [torch.randn(nb, K, P, device=dev) for _ in range(3)]      # > Infinity Cache in total
        i = [0]
        def f():
            i[0] = (i[0] + 1) % 3
            return ops.mlp_gemm(At, ring[i[0]], b, want_stats=True, pro=1, coef=coef)
        t = timed(f)
        byts = 4.0 * nb * P * (K + M)
        print("fwd %dx%d P=%6d  %7.1f us  %5.2f TB/s" % (M, K, P, t, byts / t / 1e6), flush=True)
