"""Keypoint-on-cloud nearest distances at the step's size (32 x 512 queries against 16384 points: both clouds of 16 pairs)
and the keypoint pair (16 x 512 x 512); HIP events, median."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usip_amd import ops  # noqa: E402

dev = "cuda:0"


def timed(fn, it=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(it)]
    for s, e in evs:
        s.record()
        fn()
        e.record()
    torch.cuda.synchronize()
    t = sorted(s.elapsed_time(e) for s, e in evs)
    return t[len(t) // 2] * 1e3


for B, Ma, Nb in ((16, 512, 16384), (8, 512, 512)):
    a = torch.randn(B, 3, Ma, device=dev)
    b = torch.randn(B, 3, Nb, device=dev)
    t = timed(lambda: ops.nearest(a, b))
    print("B=%d Ma=%d Nb=%d: %7.1f us" % (B, Ma, Nb, t), flush=True)
