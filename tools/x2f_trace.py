"""Where a tile of csrc/gemm_x2f.hip spends its cycles (measurement build: python tools/build_variant.py trace gemm_x2f.hip
-DUSIP_X2F_TRACE; run with USIP_LIB=tools/variants/libusip_hip_trace.so).  s_memtime stamps of workgroups 0..7:
per tile [start, stage 0 converted, loop starts, loop done, drained, next tile requested, epilogue issued]; the stage ends of
each workgroup's SECOND tile.   python tools/x2f_trace.py [M K P nb]"""
import ctypes
import os
import sys
os.environ.setdefault("USIP_ASSUME_LAUNCH_SAMPLES", "1")

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usip_amd import _lib, ops  # noqa: E402

dev = "cuda:0"
M, K, P, nb = [int(v) for v in sys.argv[1:5]] if len(sys.argv) >= 5 else (512, 512, 8192, 16)
raw = ctypes.CDLL(_lib.LIB_PATH)
ops.set_matmul_mode("f32x2")
_lib.lib().usip_set_tuning(b"gemm_split3", 2)
At = (torch.randn(K, M, device=dev) * (2.0 / K) ** 0.5)
X = torch.randn(nb, K, P, device=dev)
b = torch.randn(M, device=dev)
mu, var = X.mean(dim=(0, 2)), X.var(dim=(0, 2), unbiased=False)
istd = torch.rsqrt(var + 1e-5)
coef = torch.stack([istd, -mu * istd, mu, istd]).contiguous()
G = torch.randn(nb, M, P, device=dev)
Yg = torch.randn(nb, M, P, device=dev)
mug, istdg = Yg.mean(dim=(0, 2)), torch.rsqrt(Yg.var(dim=(0, 2), unbiased=False) + 1e-5)
cfw = torch.stack([istdg, -mug * istdg, mug, istdg]).contiguous()
c4 = ops.bn_backward_reduce(G, Yg, cfw, mug, istdg, torch.ones(M, device=dev), True)[2]
Wd = At.t().contiguous()
ops.PLANES_CACHE = {}
buf = torch.zeros(8 * 4 * 128, dtype=torch.int32, device=dev)


def timed(fn, it=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3


for what, fn in (("fwd", lambda: ops.mlp_gemm(At, X, b, want_stats=True, pro=1, coef=coef)),
                 ("dgrad", lambda: ops.mlp_gemm(Wd, G, pro=2, X2=Yg, coef=c4, tag="dgrad")) if K <= 512 else None):
    us = timed(fn)
    fn()
    raw.usip_x2f_trace_read(ctypes.c_void_p(buf.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    t = buf.cpu().numpy().astype(np.uint32).reshape(8, 4, 2, 64)
    print("%s %dx%d over %d x %d: %.1f us per launch (measurement build)" % (what, M, K, nb, P, us))
    names = ["stage0 wait+convert", "weights+barrier", "K loop", "drain+barrier", "request next", "epilogue", "(to next tile)"]
    for wg in (0, 3):
        for wave in (0, 3):
            ph = t[wg, wave, 0].astype(np.int64)
            ntile = 0
            while ntile < 8 and ph[ntile * 8 + 6] != 0:
                ntile += 1
            if ntile == 0:
                continue
            life = int((ph[(ntile - 1) * 8 + 6] - ph[0]) & 0xffffffff)
            print(" wg %d wave %d: %d tiles, %d cycles from the first tile's start to the last epilogue = %.2f GHz if it spans the launch"
                  % (wg, wave, ntile, life, life / us / 1e3))
            for k in range(ntile):
                d = [int((ph[k * 8 + j + 1] - ph[k * 8 + j]) & 0xffffffff) for j in range(6)]
                nxt = int((ph[(k + 1) * 8] - ph[k * 8 + 6]) & 0xffffffff) if k + 1 < ntile else 0
                print("   tile %d: " % k + "  ".join("%s %d" % (n, v) for n, v in zip(names, d + [nxt])))
            st = t[wg, wave, 1].astype(np.int64)
            n = K // 16
            per = [int((st[j + 1] - st[j]) & 0xffffffff) for j in range(min(n, 64) - 1)]
            if per and ntile > 1:
                print("   second tile, stage periods: mean %.0f min %d max %d | %s" % (np.mean(per), min(per), max(per),
                                                                                      " ".join(str(p) for p in per[:32])))
ops.PLANES_CACHE = None
