"""Fused narrow-layer backward (csrc/narrow_bwd.hip) vs the generic data-gradient + weight-gradient pair at the
Ball front end's shapes (B'=16 clouds, 512 nodes x 64 neighbours).  HIP events, median of 10."""
import os
os.environ.setdefault("USIP_ASSUME_LAUNCH_SAMPLES", "1")   # hand-built BatchNorm coefficients: the launch's own samples (usip_amd/ops.py::bound_covers), sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usip_amd import ops
dev = "cuda:0"
def timed(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(it)]
    for s, e in evs:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    t = sorted(s.elapsed_time(e) for s, e in evs)
    return t[len(t) // 2] * 1e3
nb, P, Cin = 16, 32768, 64
for Cout in (64, 128):
    dz = torch.randn(nb, Cout, P, device=dev); y = torch.randn(nb, Cout, P, device=dev); x = torch.randn(nb, Cin, P, device=dev)
    w2 = torch.randn(Cout, Cin, device=dev) * 0.1
    coef4 = torch.stack([1 + 0.1 * torch.randn(Cout, device=dev), 0.1 * torch.randn(Cout, device=dev), 0.05 * torch.randn(Cout, device=dev), 0.05 * torch.randn(Cout, device=dev)])
    xcoef = torch.stack([1 + 0.1 * torch.randn(Cin, device=dev), 0.1 * torch.randn(Cin, device=dev)])
    tf = timed(lambda: ops.mlp_narrow_backward(dz, y, coef4, x, xcoef, w2))
    td = timed(lambda: ops.mlp_gemm(w2, dz, pro=2, X2=y, coef=coef4, tag="dgrad"))
    tw = timed(lambda: ops.mlp_wgrad(dz, x, pro=2, G2=y, coef4=coef4, xcoef=xcoef))
    nbytes = 4.0 * nb * P * (2 * Cout + 2 * Cin)
    print("Cout=%3d: fused %6.1f us (%.2f TB/s of %d MB)   generic dgrad %6.1f + wgrad %6.1f = %6.1f us" %
          (Cout, tf, nbytes / tf / 1e6, nbytes / 1e6, td, tw, td + tw), flush=True)
