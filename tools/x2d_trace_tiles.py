"""MEASUREMENT BUILD ONLY: shader cycles between the phases of every tile of workgroups 200 / 201 of the 512 x 512 f32x2 GEMM.
    patch -p0 < tools/x2d_trace_tiles.patch && python -m usip_amd.build && python tools/x2d_trace_tiles.py [fwd|dgrad]
    patch -R -p0 < tools/x2d_trace_tiles.patch && python -m usip_amd.build
Result of round 4: profiles/r04_mfma_sustained_clock.txt, part 6."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usip_amd import _lib, ops  # noqa: E402
dev = "cuda:0"
ops.set_matmul_mode("f32x2")
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
M, K, P, nb = 512, 512, 8192, 16
At = (torch.randn(K, M, device=dev) * (2.0 / K) ** 0.5)
X = torch.randn(nb, K, P, device=dev)
b = torch.randn(M, device=dev)
mu, var = X.mean(dim=(0, 2)), X.var(dim=(0, 2), unbiased=False)
istd = torch.rsqrt(var + 1e-5)
coef = torch.stack([istd, -mu * istd, mu, istd]).contiguous()
G = torch.randn(nb, M, P, device=dev)
c4 = ops.bn_backward_reduce(G, X, coef, mu, istd, torch.ones(M, device=dev), True)[2]
Wd = At.t().contiguous()
ops.PLANES_CACHE = {}
fn = (lambda: ops.mlp_gemm(At, X, b, want_stats=True, pro=1, coef=coef)) if which == "fwd" else \
     (lambda: ops.mlp_gemm(Wd, G, pro=2, X2=X, coef=c4, tag="dgrad"))
for _ in range(4):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    fn()
e1.record()
torch.cuda.synchronize()
print("%s: %.1f us per launch" % (which, e0.elapsed_time(e1) * 100))
buf = torch.zeros(2 * 4 * 64, dtype=torch.int32, device=dev)
lib = ctypes.CDLL(_lib.lib()._name)
lib.usip_x2d_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
lib.usip_x2d_trace_read(buf.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
t = (buf.cpu().numpy().astype(np.int64) & 0xffffffff).reshape(2, 4, 64)
names = ["loads issued", "X(0) arrived+converted", "DMA landed, barrier", "K loop", "drain + barrier", "epilogue issued", "tile-end barrier", "(next tile setup)"]
for wg in range(2):
    for w in (0, 3):
        r = t[wg, w, :32].reshape(4, 8)
        print("wg %d wave %d: tile lengths %s" % (wg, w, [int(r[i + 1, 0] - r[i, 0]) for i in range(3)]))
        for i in range(4):
            d = [int(r[i, j + 1] - r[i, j]) for j in range(7)] + ([int(r[i + 1, 0] - r[i, 7])] if i < 3 else [0])
            print("   tile %d: " % i + "  ".join("%s %d" % (n, v) for n, v in zip(names, d)))
