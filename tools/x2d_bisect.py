"""Which epilogue option breaks the direct kernel's forward?"""
import os
os.environ.setdefault("USIP_ASSUME_LAUNCH_SAMPLES", "1")   # hand-built BatchNorm coefficients: the launch's own samples (usip_amd/ops.py::bound_covers), sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usip_amd import _lib, ops  # noqa: E402
dev = "cuda:0"
ops.set_matmul_mode("f32x2")
_lib.lib().usip_set_tuning(b"gemm_split3", 2)
nb, K, M, P = 4, 256, 256, 8192
torch.manual_seed(0)
At = (torch.randn(K, M, device=dev) * (2.0 / K) ** 0.5)
X = torch.randn(nb, K, P, device=dev)
b = torch.randn(M, device=dev)
mu, var = X.mean(dim=(0, 2)), X.var(dim=(0, 2), unbiased=False)
istd = torch.rsqrt(var + 1e-5)
coef = torch.stack([istd, -mu * istd, mu, istd]).contiguous()
for bias in (None, b):
    for st in (False, True):
        _lib.lib().usip_set_tuning(b"x2_direct", 1)
        ref = ops.mlp_gemm(At, X, bias, want_stats=st, pro=1, coef=coef)
        _lib.lib().usip_set_tuning(b"x2_direct", 0)
        y = ops.mlp_gemm(At, X, bias, want_stats=st, pro=1, coef=coef)
        d = (y[0] - ref[0]).abs()
        bad = (d > 1e-5 * ref[0].abs().max())
        print("bias %s stats %s: max diff %.2e  bad frac %.4f" % (bias is not None, st, float(d.max() / ref[0].abs().max()), float(bad.float().mean())),
              "stats diff %.2e" % (float((y[1] - ref[1]).abs().max() / ref[1].abs().max()) if st else 0.0))
        if bad.any():
            idx = bad.nonzero()[:8].tolist()
            print("   first bad (b, ch, pos):", idx)
            ch = bad.any(dim=2).any(dim=0).nonzero().flatten()
            ps = bad.any(dim=1).any(dim=0).nonzero().flatten()
            print("   bad channels: %d (first %s)  bad positions %% 128: %s" % (ch.numel(), ch[:12].tolist(), sorted(set((ps % 128).tolist()))[:40]))
_lib.lib().usip_set_tuning(b"x2_direct", 1)
ref = ops.mlp_gemm(At, X, None, want_stats=True, pro=1, coef=coef)[0]
_lib.lib().usip_set_tuning(b"x2_direct", 0)
y = ops.mlp_gemm(At, X, None, want_stats=True, pro=1, coef=coef)[0]
torch.set_printoptions(precision=4, linewidth=200)
print("ref[0, :4, :12]\n", ref[0, :4, :12])
print("y  [0, :4, :12]\n", y[0, :4, :12])
# is y a permutation of ref within the 32 x 32 block?
blk_r, blk_y = ref[0, :32, :32], y[0, :32, :32]
for r in range(2):
    for cidx in range(3):
        m = (blk_r == blk_y[r, cidx]).nonzero()
        print("y[%d,%d] = %.4f found in ref block at" % (r, cidx, float(blk_y[r, cidx])), m.tolist()[:3])
