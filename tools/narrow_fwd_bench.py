"""Streaming narrow forward kernel vs the generic tile kernel (same launch through ops.mlp_gemm), ring of inputs
larger than the Infinity Cache."""
import os
import sys
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
for (M, K, P, rb) in [(64, 64, 32768, False), (128, 64, 32768, False)]:
    At = torch.randn(K, M, device=dev) * 0.1
    b = torch.randn(M, device=dev)
    coef = torch.stack([1 + 0.1 * torch.randn(K, device=dev), 0.1 * torch.randn(K, device=dev)]).contiguous()
    ring = [torch.randn(nb, K, P, device=dev) for _ in range(3)]
    rowbias = torch.randn(nb, M, P // 64, device=dev) if rb else None
    i = [0]
    PROV = int(os.environ.get("PROV", "1")); STATS = bool(int(os.environ.get("STATS", "1")))
    def f():
        i[0] = (i[0] + 1) % 3
        return ops.mlp_gemm(At, ring[i[0]], b, want_stats=STATS, pro=PROV, coef=coef if PROV else None, rowbias=rowbias, rb_group=64 if rb else 1)
    byts = 4.0 * nb * P * (K + M)
    res = []
    for streaming in (False, True):
        ops.NARROW_FWD = streaming
        t = timed(f)
        res.append("%s %7.1f us %5.2f TB/s" % ("streaming" if streaming else "generic  ", t, byts / t / 1e6))
    print("fwd %3dx%d P=%5d rowbias=%d   %s" % (M, K, P, rb, "   ".join(res)), flush=True)
