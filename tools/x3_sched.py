import os
os.environ.setdefault("USIP_ASSUME_LAUNCH_SAMPLES", "1")   # hand-built BatchNorm coefficients: the launch's own samples (usip_amd/ops.py::bound_covers), sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usip_amd import _lib, ops
dev = "cuda:0"
ops.set_matmul_mode("f32x3")
lib = _lib.lib()
lib.usip_set_tuning(b"gemm_split3", 2)
def timed(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(it)]
    for s, e in evs:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    t = sorted(s.elapsed_time(e) for s, e in evs)
    return t[len(t) // 2] * 1e3
for (M, K, P, nb) in [(512, 512, 8192, 16), (256, 256, 8192, 16), (256, 131, 8192, 16), (128, 128, 32768, 16)]:
    At = torch.randn(K, M, device=dev) * (2.0 / K) ** 0.5
    X = torch.randn(nb, K, P, device=dev); X2 = torch.randn(nb, K, P, device=dev)
    b = torch.randn(M, device=dev)
    coef = torch.stack([1 + 0.1 * torch.randn(K, device=dev), 0.1 * torch.randn(K, device=dev), 0.05 * torch.randn(K, device=dev), 0.05 * torch.randn(K, device=dev)])
    ops.PLANES_CACHE = {}
    truth = torch.matmul(At.double().t().unsqueeze(0), X[:1].double()) + b.double().view(1, M, 1)
    for name, bits in [("256x128 tile", 0), ("128x128 tile", 1), ("256x256 (no stats)", 4)]:
        lib.usip_set_tuning(b"x3_gemm_tile", bits)
        ops.PLANES_CACHE = {}
        t0 = timed(lambda: ops.mlp_gemm(At, X, b, want_stats=True))
        t1 = timed(lambda: ops.mlp_gemm(At, X, b, want_stats=True, pro=1, coef=coef[:2].contiguous()))
        t2 = timed(lambda: ops.mlp_gemm(At, X, pro=2, X2=X2, coef=coef))
        err = float((ops.mlp_gemm(At, X[:1].contiguous(), b)[0].double() - truth).abs().max() / truth.abs().max())
        fl = 2.0 * M * K * P * nb
        print("M=%d K=%d %-18s fwd %6.1f us (%5.1f TF)  fwd+bnrelu %6.1f us (%5.1f TF)  dgrad(bn-bwd) %6.1f us (%5.1f TF)  err %.1e" %
              (M, K, name, t0, fl / t0 / 1e6, t1, fl / t1 / 1e6, t2, fl / t2 / 1e6, err), flush=True)
lib.usip_set_tuning(b"x3_gemm_tile", 0)
