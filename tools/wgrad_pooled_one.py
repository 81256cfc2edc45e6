"""A few launches of the POOLED f32x2 weight gradient (512 x 512: dZ = (k == arg) ? dpooled : 0 synthesised in the
prologue, BN+ReLU on X) for counter passes and timing:  python tools/wgrad_pooled_one.py [M N P nb [knob value]]"""
import os
os.environ.setdefault("USIP_ASSUME_LAUNCH_SAMPLES", "1")   # hand-built BatchNorm coefficients: the launch's own samples
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usip_amd import ops  # noqa: E402

dev = "cuda:0"
ops.set_matmul_mode("f32x2")
M, N, P, nb = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (512, 512, 8192, 16)
K = 16
if len(sys.argv) > 5:                        # a library knob: name value
    from usip_amd import _lib
    _lib.lib().usip_set_tuning(sys.argv[5].encode(), int(sys.argv[6]))
dp = torch.randn(nb, M, P // K, device=dev)
arg = torch.randint(0, K, (nb, M, P // K), device=dev, dtype=torch.int32)
Y = torch.randn(nb, M, P, device=dev)
X = torch.randn(nb, N, P, device=dev)
coef4 = torch.cat([torch.rand(4, M, device=dev) + 0.5, torch.full((1, M), 6.0, device=dev)]).contiguous()
mu, var = X.mean(dim=(0, 2)), X.var(dim=(0, 2), unbiased=False)
istd = torch.rsqrt(var + 1e-5)
xcoef = torch.stack([istd, -mu * istd, mu, istd]).contiguous()


def call():
    return ops.mlp_wgrad(None, X, pro=3, G2=Y, coef4=coef4, xcoef=xcoef, pool=(dp, arg, K))


for _ in range(3):
    call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(6):
    call()
e1.record()
torch.cuda.synchronize()
print("pooled wgrad %dx%d P=%d nb=%d: %.1f us per call (kernel + reduce)" % (M, N, P, nb, e0.elapsed_time(e1) / 6 * 1e3))
