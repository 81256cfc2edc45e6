"""A/B of the f32x2 data-gradient GEMM tile (x3_gemm_tile knob: 0 = 256 x 128, 4 waves, two workgroups per CU;
5 = 256 x 256, 8 waves, one workgroup per CU) at the step's wide-layer shapes."""
import os
os.environ.setdefault("USIP_ASSUME_LAUNCH_SAMPLES", "1")   # hand-built BatchNorm coefficients: the launch's own samples (usip_amd/ops.py::bound_covers)
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usip_amd import _lib, ops  # noqa: E402

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


ops.set_matmul_mode(sys.argv[1] if len(sys.argv) > 1 else "f32x2")
_lib.lib().usip_set_tuning(b"gemm_split3", 2)
for (M, K, P, nb) in [(512, 512, 8192, 16), (512, 256, 8192, 16), (256, 256, 8192, 16)]:
    At = (torch.randn(K, M, device=dev) * (2.0 / K) ** 0.5)
    G = torch.randn(nb, M, P, device=dev)
    Yg = torch.randn(nb, M, P, device=dev)
    mug, istdg = Yg.mean(dim=(0, 2)), torch.rsqrt(Yg.var(dim=(0, 2), unbiased=False) + 1e-5)
    cfw = torch.stack([istdg, -mug * istdg, mug, istdg]).contiguous()
    c4 = ops.bn_backward_reduce(G, Yg, cfw, mug, istdg, torch.ones(M, device=dev), True)[2]
    Wd = At.t().contiguous()
    ref = None
    for rnd in range(2):
        for knob in (0, 5):
            _lib.lib().usip_set_tuning(b"x3_gemm_tile", knob)
            ops.PLANES_CACHE = {}
            t2 = timed(lambda: ops.mlp_gemm(Wd, G, pro=2, X2=Yg, coef=c4, tag="dgrad"))
            y = ops.mlp_gemm(Wd, G, pro=2, X2=Yg, coef=c4, tag="dgrad")[0]
            if ref is None:
                ref = y.clone()
            print("dgrad out %d x in %d  tile knob %d: %7.1f us  max diff vs knob 0 %.1e" % (
                K, M, knob, t2, float((y - ref).abs().max() / ref.abs().max())), flush=True)
    ops.PLANES_CACHE = None
_lib.lib().usip_set_tuning(b"x3_gemm_tile", 0)
