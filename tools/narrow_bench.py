"""Which feature of the narrow forward GEMMs costs what (conv4 of the ball front end: 128 out, 64 in)."""
import os
os.environ.setdefault("USIP_ASSUME_LAUNCH_SAMPLES", "1")   # hand-built BatchNorm coefficients: the launch's own samples (usip_amd/ops.py::bound_covers), sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usip_amd import ops


def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3


dev = "cuda:0"
nb, Kn, Mn = 16, 64, 512
P = Mn * Kn
for (M, K) in [(128, 64), (64, 64), (128, 128), (64, 7)]:
    At = torch.randn(K, M, device=dev); X = torch.randn(nb, K, P, device=dev); b = torch.randn(M, device=dev)
    coef = torch.rand(4, K, device=dev)
    rb = torch.randn(nb, M, Mn, device=dev)
    mb = 4.0 * nb * P * (K + M) / 1e6
    for name, kw in [("plain", dict()), ("stats", dict(want_stats=True)), ("pro1", dict(pro=1, coef=coef)),
                     ("pro1+stats", dict(pro=1, coef=coef, want_stats=True)),
                     ("pro1+stats+rowbias", dict(pro=1, coef=coef, want_stats=True, rowbias=rb, rb_group=Kn))]:
        us = t(lambda: ops.mlp_gemm(At, X, b, **kw))
        print("fwd M=%3d K=%3d %-20s %7.1f us  %5.2f TB/s  %5.1f TF" % (M, K, name, us, mb / us, 2.0 * M * K * P * nb / us / 1e6))
