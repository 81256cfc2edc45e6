"""The shader clock the chip holds while a given kernel of the step runs (tools/probes/clockmon.hip: a one-wave monitor
samples s_memtime / s_memrealtime on a side stream while the kernel is launched back to back on torch's stream).
    python tools/clock_under_kernel.py            -> one line per kernel + the replayed step
CAVEAT (measured, round 4): the monitor wave takes a wave slot.  Kernels that fill the register file with persistent
workgroups (gemm_x2d: 2 x 4 waves of 256 registers per CU) lose one workgroup slot on the CU the monitor sits on; the
displaced workgroup runs after the others (226 -> 305 us) and the chip idles, at a high clock, meanwhile: for those
kernels the number printed here is too high and the in-kernel probe (tools/x2d_trace.patch: 1.07 GHz) is the
measurement.  For the streaming kernels (small workgroups, many of them) the monitor costs nothing."""
import ctypes
import os
os.environ.setdefault("USIP_ASSUME_LAUNCH_SAMPLES", "1")   # hand-built BatchNorm coefficients: the launch's own samples (usip_amd/ops.py::bound_covers)
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from usip_amd import ops  # noqa: E402

SO = os.path.join(ROOT, "tools", "probes", "libclockmon.so")
if not os.path.exists(SO):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC",
                           os.path.join(ROOT, "tools", "probes", "clockmon.hip"), "-o", SO])
mon = ctypes.CDLL(SO)
mon.clockmon_start.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = "cuda:0"
MAXS = 1 << 20
buf = torch.zeros(2 * MAXS, dtype=torch.int64, device=dev)


def clock_during(fn, reps, capture=True):
    """(median GHz over 100-us windows, min, max, us per call) while fn runs reps times back to back"""
    for _ in range(3):
        fn()
    torch.cuda.current_stream().synchronize()
    g = None
    if capture:                                              # back to back on the GPU: the host cannot launch these fast enough
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(10):
                    fn()
        torch.cuda.current_stream().wait_stream(s)
        g.replay()
        torch.cuda.current_stream().synchronize()
    rc = mon.clockmon_start(buf.data_ptr(), MAXS)
    assert rc == 0, rc
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    if g is not None:
        for _ in range(max(1, reps // 10)):
            g.replay()
        reps = max(1, reps // 10) * 10
    else:
        for _ in range(reps):
            fn()
    e1.record()
    torch.cuda.current_stream().synchronize()
    n = mon.clockmon_stop()
    assert n > 100, n
    t = buf[:2 * n].cpu().numpy().reshape(n, 2)
    cyc, real = t[:, 0].astype(np.float64), t[:, 1].astype(np.float64)
    # drop the first and last 15 % (ramp), 100-us windows = 10 000 ticks of the 100 MHz counter
    lo, hi = int(0.15 * n), int(0.85 * n)
    cyc, real = cyc[lo:hi], real[lo:hi]
    w = ((real - real[0]) // 10000).astype(np.int64)
    ghz = []
    for k in np.unique(w):
        m = np.nonzero(w == k)[0]
        if len(m) > 4:
            ghz.append((cyc[m[-1]] - cyc[m[0]]) / ((real[m[-1]] - real[m[0]]) * 10.0))
    ghz = np.array(ghz)
    return float(np.median(ghz)), float(ghz.min()), float(ghz.max()), e0.elapsed_time(e1) * 1e3 / reps


def bn_inputs(nb, C, P):
    x = torch.randn(nb, C, P, device=dev)
    mu, var = x.mean(dim=(0, 2)), x.var(dim=(0, 2), unbiased=False)
    istd = torch.rsqrt(var + 1e-5)
    return x, torch.stack([istd, -mu * istd, mu, istd]).contiguous(), mu, istd


def main():
    ops.set_matmul_mode("f32x2")
    rows = []
    big = torch.randn(64 * 1024 * 1024, device=dev)
    out = torch.empty_like(big)
    rows.append(("element-wise pass, 268 MB read + 268 MB written (ATen mul)", lambda: torch.mul(big, 2.0, out=out), 40))
    M, K, P, nb = 512, 512, 8192, 16
    At = torch.randn(K, M, device=dev) * (2.0 / K) ** 0.5
    X, coef, _, _ = bn_inputs(nb, K, P)
    b = torch.randn(M, device=dev)
    ops.PLANES_CACHE = {}
    rows.append(("gemm_x2d forward 512 x 512 (f32x2)", lambda: ops.mlp_gemm(At, X, b, want_stats=True, pro=1, coef=coef), 60))
    G = torch.randn(nb, M, P, device=dev)
    Y, cfw, mug, istdg = bn_inputs(nb, M, P)
    c4 = ops.bn_backward_reduce(G, Y, cfw, mug, istdg, torch.ones(M, device=dev), True)[2]
    Wd = At.t().contiguous()
    rows.append(("gemm_x2d data gradient 512 x 512", lambda: ops.mlp_gemm(Wd, G, pro=2, X2=Y, coef=c4, tag="dgrad"), 60))
    rows.append(("wgrad_x3<2> weight gradient 512 x 512", lambda: ops.mlp_wgrad(G, X, pro=2, G2=Y, coef4=c4, xcoef=coef), 50))
    rows.append(("bn_backward_reduce 512 channels", lambda: ops.bn_backward_reduce(G, Y, cfw, mug, istdg, torch.ones(M, device=dev), True), 120))
    ops.set_matmul_mode("f32")
    rows.append(("gemm_kernel forward 512 x 512 (fp32 MFMA)", lambda: ops.mlp_gemm(At, X, b, want_stats=True, pro=1, coef=coef), 30))
    ops.set_matmul_mode("f32x2")
    X64, coef64, _, _ = bn_inputs(16, 64, 32768)
    A64 = torch.randn(64, 64, device=dev) * 0.2
    b64 = torch.randn(64, device=dev)
    rows.append(("narrow_fwd 64 -> 64 (fp32 MFMA, streaming)", lambda: ops.mlp_gemm(A64, X64, b64, want_stats=True, pro=1, coef=coef64), 150))
    X128, coef128, _, _ = bn_inputs(16, 128, 32768)
    A128 = torch.randn(128, 128, device=dev) * 0.1
    b128 = torch.randn(128, device=dev)
    rows.append(("gemm_x2r forward 128 x 128 (f32x2, streaming)", lambda: ops.mlp_gemm(A128, X128, b128, want_stats=True, pro=1, coef=coef128), 100))
    print("%-62s %8s %8s %8s %10s" % ("kernel", "GHz med", "min", "max", "us / call"))
    for name, fn, reps in rows:
        med, lo, hi, us = clock_during(fn, reps)
        print("%-62s %8.2f %8.2f %8.2f %10.1f" % (name, med, lo, hi, us), flush=True)
    # the replayed training step
    from usip_amd import synth
    from usip_amd.networks import DetectorOptions
    from usip_amd.step import DetectorStep, batch_to_device
    ops.PLANES_CACHE = None
    st = DetectorStep("ball", DetectorOptions(surface_normal_len=4, node_knn_k_1=16), torch.device(dev), with_optimizer=True, graph=True)
    batch = batch_to_device(synth.make_pair_batch(1234, 8, 16384, 512, 4, "slab"), torch.device(dev))
    for _ in range(6):
        st.step(batch)
    med, lo, hi, us = clock_during(lambda: st.step(batch), 200, capture=False)
    print("%-62s %8.2f %8.2f %8.2f %10.1f" % ("the replayed training step (RPN_Detector_Ball, f32x2)", med, lo, hi, us))


if __name__ == "__main__":
    main()
