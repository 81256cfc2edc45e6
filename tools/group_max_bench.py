"""group_max_act (fused BN + ReLU + max over K, csrc/group.hip) at the step's four pooling shapes; HIP events, median."""
import os
os.environ.setdefault("USIP_ASSUME_LAUNCH_SAMPLES", "1")   # hand-built BatchNorm coefficients: the launch's own samples (usip_amd/ops.py::bound_covers)
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usip_amd import ops  # noqa: E402

dev = "cuda:0"


def timed(fn, it=20):
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


for (B, C, M, K) in [(16, 64, 512, 64), (16, 128, 512, 64), (16, 256, 512, 16), (16, 512, 512, 16)]:
    ring = [torch.randn(B, C, M, K, device=dev) for _ in range(3)]
    coef = torch.stack([1 + 0.1 * torch.randn(C, device=dev), 0.1 * torch.randn(C, device=dev)])
    i = [0]

    def run():
        i[0] += 1
        return ops.group_max_act(ring[i[0] % 3], coef, True, want_yarg=True)
    t = timed(run)
    y = ring[0]
    pooled, arg, yarg = ops.group_max_act(y, coef, True, want_yarg=True)
    act = torch.relu(y * coef[0].view(1, C, 1, 1) + coef[1].view(1, C, 1, 1))
    ref, refarg = act.max(dim=3)
    ok = bool(torch.equal(arg.long(), refarg)) and float((pooled - ref).abs().max()) < 1e-5
    nbytes = 4.0 * B * C * M * (K + 3)
    print("C=%3d K=%2d: %6.1f us  %5.0f GB/s  matches torch.max: %s" % (C, K, t, nbytes / t / 1e3, ok), flush=True)
