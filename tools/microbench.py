"""Per-kernel timing of the native operators at BASELINE config 3 sizes (HIP events on the
launch stream).  Usage: python tools/microbench.py [--bprime 16]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usip_amd import ops, synth  # noqa: E402


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for s, e in evs:
        s.record()
        fn()
        e.record()
    torch.cuda.synchronize()
    t = sorted(s.elapsed_time(e) for s, e in evs)
    return t[len(t) // 2] * 1e-3, t[0] * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bprime", type=int, default=16)
    ap.add_argument("--n", type=int, default=16384)
    ap.add_argument("--m", type=int, default=512)
    ap.add_argument("--k", type=int, default=64)
    a = ap.parse_args()
    dev = "cuda:0"
    B, N, M, K = a.bprime, a.n, a.m, a.k
    rng = np.random.default_rng(0)
    print("device:", torch.cuda.get_device_name(0), " B'=%d N=%d M=%d K=%d" % (B, N, M, K))
    for kind in ("cube", "slab"):
        x = torch.from_numpy(np.stack([synth.make_cloud(rng, N, kind) for _ in range(B)])).to(dev)
        node = x[:, :, :M].contiguous()
        dist = ops.pairwise_dist(node, x)
        t, tmin = timeit(lambda: ops.pairwise_dist(node, x))
        print("%-5s pairwise_dist       %8.1f us (min %.1f)  %7.1f GB/s written" % (kind, t * 1e6, tmin * 1e6, 4 * B * M * N / t / 1e9))
        inside = (dist <= 2.0)
        cs = torch.cumsum(inside.int(), -1)
        kth = (cs >= K).int().argmax(-1)
        prefix = torch.where(cs[..., -1] >= K, kth + 1, torch.full_like(kth, N)).sum().item()
        alg = 4 * prefix + 4 * B * M * K
        t, tmin = timeit(lambda: ops.ball_query(dist, 2.0, K))
        print("%-5s ball_query(dist-in) %8.1f us (min %.1f)  alg %.2f MB -> %7.1f GB/s = %.1f%% of 8 TB/s" %
              (kind, t * 1e6, tmin * 1e6, alg / 1e6, alg / t / 1e9, alg / t / 8e12 * 100))
        t, tmin = timeit(lambda: ops.ball_query_coords(node, x, 2.0, K))
        print("%-5s ball_query_coords   %8.1f us (min %.1f)  %.1f Gpair/s" % (kind, t * 1e6, tmin * 1e6, B * M * N / t / 1e9))
    for C in (64, 128):
        data = torch.randn(B, C, N, device=dev)
        idx = torch.randint(0, M, (B, N), device=dev, dtype=torch.int32)
        t, tmin = timeit(lambda: ops.index_max(data, idx, M))
        alg = 4 * B * C * N + 4 * B * N + 4 * B * C * M
        print("index_max C=%-3d          %8.1f us (min %.1f)  alg %.2f MB -> %7.1f GB/s = %.1f%% of 8 TB/s" %
              (C, t * 1e6, tmin * 1e6, alg / 1e6, alg / t / 1e9, alg / t / 8e12 * 100))


if __name__ == "__main__":
    main()
