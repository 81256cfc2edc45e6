"""Where does the direct f32x2 GEMM differ from the LDS-staged one?  Per (cloud, 256-row tile, 128-position tile)."""
import os
os.environ.setdefault("USIP_ASSUME_LAUNCH_SAMPLES", "1")   # hand-built BatchNorm coefficients: the launch's own samples (usip_amd/ops.py::bound_covers)
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usip_amd import _lib, ops  # noqa: E402

dev = "cuda:0"
ops.set_matmul_mode("f32x2")
_lib.lib().usip_set_tuning(b"gemm_split3", 2)
nb, K, M, P = [int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (16, 512, 512, 8192))]
knob = int(sys.argv[5]) if len(sys.argv) > 5 else 0
torch.manual_seed(0)
At = (torch.randn(K, M, device=dev) * (2.0 / K) ** 0.5)
X = torch.randn(nb, K, P, device=dev)
b = torch.randn(M, device=dev)
mu, var = X.mean(dim=(0, 2)), X.var(dim=(0, 2), unbiased=False)
istd = torch.rsqrt(var + 1e-5)
coef = torch.stack([istd, -mu * istd, mu, istd]).contiguous()
ops.PLANES_CACHE = {}
_lib.lib().usip_set_tuning(b"x2_direct", 1)
ref = ops.mlp_gemm(At, X, b, want_stats=True, pro=1, coef=coef)[0]
_lib.lib().usip_set_tuning(b"x2_direct", knob)
for rep in range(3):
    y = ops.mlp_gemm(At, X, b, want_stats=True, pro=1, coef=coef)[0]
    d = (y - ref).abs() / ref.abs().max()
    Mp, Pp = (M + 255) // 256 * 256, (P + 127) // 128 * 128
    dd = torch.zeros(nb, Mp, Pp, device=dev)
    dd[:, :M, :P] = d
    t = dd.view(nb, Mp // 256, 256, Pp // 128, 128)
    tile = t.amax(dim=(2, 4))                                  # [nb, mt, pt]
    bad = (tile > 1e-5)
    print("rep %d: max diff %.2e, bad tiles %d of %d" % (rep, float(d.max()), int(bad.sum()), bad.numel()))
    if bad.any():
        idx = bad.nonzero()[:12].tolist()
        print("   first bad (b, mt, pt):", idx)
        bi, mi, pi = idx[0]
        one = t[bi, mi, :, pi, :]                              # [256 ch, 128 pos]
        print("   in that tile: bad per wave (32 positions):", [int((one[:, w * 32:(w + 1) * 32] > 1e-5).sum()) for w in range(4)],
              " bad per channel tile (32 rows):", [int((one[r * 32:(r + 1) * 32] > 1e-5).sum()) for r in range(8)])
        bl = (one > 1e-5).nonzero().tolist()
        print("   bad (row, pos) in tile:", bl[:40])
        sub = one[160:192]                                     # [32 ch, 128 pos]
        rows = (sub > 1e-5).any(dim=1).nonzero().flatten().tolist()
        cols = (sub > 1e-5).any(dim=0).nonzero().flatten().tolist()
        print("   bad channels (of 160..191):", rows, " bad positions:", cols)
        yy = y[bi, mi * 256 + 160:mi * 256 + 192, pi * 128:(pi + 1) * 128]
        rr = ref[bi, mi * 256 + 160:mi * 256 + 192, pi * 128:(pi + 1) * 128]
        if cols:
            c0 = cols[0]
            print("   y   :", [round(float(v), 4) for v in yy[:6, c0]])
            print("   ref :", [round(float(v), 4) for v in rr[:6, c0]])
        print("   bad tiles per cloud:", bad.sum(dim=(1, 2)).tolist())
        print("   bad tiles per pt %% 8:", [int(bad[:, :, r::8].sum()) for r in range(8)])
