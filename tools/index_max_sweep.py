"""index_max launch-geometry sweep (channel rows per workgroup x prefetch depth) at BASELINE configs[2] sizes, on a
ring of distinct inputs larger than the 256 MB Infinity Cache.  HIP events on the launch stream, median of 12.

    python tools/index_max_sweep.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usip_amd import _lib, ops  # noqa: E402

dev = "cuda:0"
B, N, M = 16, 16384, 512
lib = _lib.lib()
print("device:", torch.cuda.get_device_name(0))
for C in (64, 128):
    n_ring = 8 if C == 64 else 4
    data = [torch.randn(B, C, N, device=dev) for _ in range(n_ring)]
    idx = [torch.randint(0, M, (B, N), device=dev, dtype=torch.int32) for _ in range(n_ring)]
    alg = 4.0 * (B * C * N + B * N + B * C * M)
    want = ops.index_max(data[0], idx[0], M)
    if True:
      for ch, u, th in [(0, 0, 0)] + [(c, uu, tt) for tt in (256, 1024) for c in (1, 2, 4) for uu in (1, 2)]:
            lib.usip_set_tuning(b"index_max_ch", ch)
            lib.usip_set_tuning(b"index_max_unroll", u)
            lib.usip_set_tuning(b"index_max_threads", th)
            assert torch.equal(ops.index_max(data[0], idx[0], M), want)
            for i in range(3):
                ops.index_max(data[i % n_ring], idx[i % n_ring], M)
            torch.cuda.synchronize()
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(12)]
            for i, (s, e) in enumerate(evs):
                s.record()
                ops.index_max(data[i % n_ring], idx[i % n_ring], M)
                e.record()
            torch.cuda.synchronize()
            t = sorted(s.elapsed_time(e) for s, e in evs)
            med = t[len(t) // 2] * 1e-3
            g = ops.index_max_geometry(B, C, N, M)
            print("C=%-3d ch=%d u=%d t=%d (runs as %s, %4d workgroups): %6.1f us (min %.1f)  %6.0f GB/s = %4.1f%% of 8 TB/s"
                  % (C, ch, u, th, g, B * C // g[0], med * 1e6, t[0] * 1e3, alg / med / 1e9, alg / med / 8e12 * 100), flush=True)
    lib.usip_set_tuning(b"index_max_ch", 0)
    lib.usip_set_tuning(b"index_max_unroll", 0)
    lib.usip_set_tuning(b"index_max_threads", 0)
    del data, idx
