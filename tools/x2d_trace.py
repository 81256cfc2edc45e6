"""MEASUREMENT BUILD ONLY: per-stage timeline of workgroups 200 / 201 of the 512 x 512 f32x2 forward GEMM.
    patch -p0 < tools/x2d_trace.patch && python -m usip_amd.build && python tools/x2d_trace.py ; patch -R -p0 < tools/x2d_trace.patch
The patch puts s_memtime (= shader cycles, tools/probes/memtime_rate.hip) in front of the counted wait, behind it and behind
the barrier of every stage (values kept in the lanes of one VGPR with v_writelane: no LDS, no branches in the loop) and
exports usip_x2d_trace_read.  Result of round 4: profiles/r04_mfma_sustained_clock.txt, part 5."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usip_amd import _lib, ops  # noqa: E402

dev = "cuda:0"
ops.set_matmul_mode("f32x2")
M, K, P, nb = 512, 512, 8192, 16
At = (torch.randn(K, M, device=dev) * (2.0 / K) ** 0.5)
X = torch.randn(nb, K, P, device=dev)
b = torch.randn(M, device=dev)
mu, var = X.mean(dim=(0, 2)), X.var(dim=(0, 2), unbiased=False)
istd = torch.rsqrt(var + 1e-5)
coef = torch.stack([istd, -mu * istd, mu, istd]).contiguous()
ops.PLANES_CACHE = {}
for _ in range(4):
    ops.mlp_gemm(At, X, b, want_stats=True, pro=1, coef=coef)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.mlp_gemm(At, X, b, want_stats=True, pro=1, coef=coef)
e1.record()
torch.cuda.synchronize()
print("kernel: %.1f us per launch" % (e0.elapsed_time(e1) * 100))
buf = torch.zeros(2 * 4 * 64, dtype=torch.int32, device=dev)
lib = ctypes.CDLL(_lib.lib()._name)
lib.usip_x2d_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
lib.usip_x2d_trace_read(buf.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
raw = buf.cpu().numpy().astype(np.int64).reshape(2, 4, 64) & 0xffffffff
print('ticks from the start of a workgroup to the end of its last tile:', raw[:, :, 63].tolist())
t = raw[:, :, :63].reshape(2, 4, 21, 3) & 0xffffffff
for wg in range(2):
    for w in range(4):
        r = t[wg, w]
        period = np.diff(r[:, 0])
        wait = r[:, 1] - r[:, 0]
        bar = r[:, 2] - r[:, 1]
        print("wg %d wave %d: stage period mean %.0f (min %d max %d) | counted wait %.0f (max %d) | barrier %.0f (max %d) | rest %.0f"
              % (wg, w, period.mean(), period.min(), period.max(), wait.mean(), wait.max(), bar.mean(), bar.max(),
                 period.mean() - wait.mean() - bar.mean()))
    r = t[wg, 0]
    print("  wave 0, stages 4..23: period / counted wait / barrier")
    print("   " + "  ".join("%d/%d/%d" % (r[k + 1, 0] - r[k, 0], r[k, 1] - r[k, 0], r[k, 2] - r[k, 1]) for k in range(20)))
