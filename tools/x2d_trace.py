"""MEASUREMENT BUILD ONLY: per-stage timeline of workgroups 200 / 201 of the 512 x 512 f32x2 forward GEMM.
    patch -p0 < tools/x2d_trace.patch && python -m usip_amd.build && python tools/x2d_trace.py ; patch -R -p0 < tools/x2d_trace.patch
The patch puts s_memtime (= shader cycles, tools/probes/memtime_rate.hip) in front of the counted wait, behind it and behind
the barrier of every stage (values kept in the lanes of one VGPR with v_writelane: no LDS, no branches in the loop) and
exports usip_x2d_trace_read; it also adds the loop-ablation knobs (x2_direct = 64: no weight DMA, 128: no operand loads).
    python tools/x2d_trace.py 0 64 128 192 32 96      (knob values; two rounds in one process)
Result of round 4: profiles/r04_mfma_sustained_clock.txt, parts 5 and 7."""
import ctypes
import os
os.environ.setdefault("USIP_ASSUME_LAUNCH_SAMPLES", "1")   # hand-built BatchNorm coefficients: the launch's own samples (usip_amd/ops.py::bound_covers)
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
buf = torch.zeros(2 * 4 * 64, dtype=torch.int32, device=dev)
lib = ctypes.CDLL(_lib.lib()._name)
lib.usip_x2d_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
knobs = [int(v) for v in sys.argv[1:]] or [0]
for rnd in range(2):
    for kn in knobs:
        _lib.lib().usip_set_tuning(b"x2_direct", kn)
        for _ in range(4):
            ops.mlp_gemm(At, X, b, want_stats=True, pro=1, coef=coef)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.mlp_gemm(At, X, b, want_stats=True, pro=1, coef=coef)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 50
        lib.usip_x2d_trace_read(buf.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        raw = buf.cpu().numpy().astype(np.int64).reshape(2, 4, 64) & 0xffffffff
        life = raw[:, :, 63].mean()
        t = raw[:, :, :63].reshape(2, 4, 21, 3)
        period = np.diff(t[:, :, :, 0], axis=2)
        bar = t[:, :, :, 2] - t[:, :, :, 1]
        wait = t[:, :, :, 1] - t[:, :, :, 0]
        print("knob %4d: %6.1f us per launch | workgroup lifetime %7.0f cycles -> %.2f GHz | stage period mean %5.0f min %5d max %5d | counted wait %4.0f (max %d) | barrier %4.0f (per wave: %s)"
              % (kn, us, life, life / (us * 1e3), period.mean(), period.min(), period.max(), wait.mean(), wait.max(), bar.mean(),
                 " ".join("%d" % v for v in bar.mean(axis=2).reshape(-1))), flush=True)
_lib.lib().usip_set_tuning(b"x2_direct", 0)
