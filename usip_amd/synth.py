"""Synthetic inputs for the USIP detector hot path (there are no datasets here).

Cloud generators follow the shapes the reference's loaders hand to the detector
(data/kitti_detector_loader.py:144 -> pc 3xN, sn CsxN, node 3xM per cloud, plus a second
view and the rigid transform that links them, data/augmentation.py:199-248):

  slab   KITTI/Oxford-like, camera frame, y vertical: x,z ~ U(-50,50), y ~ N(0,1)
  cube   x,y,z ~ U(-50,50): ~0.5 points per r=2 ball, every ball_query row is a full scan
         (the defining case for the ball_query HBM roofline)
  sphere ModelNet-like: points on the unit sphere surface scaled by U(0.8,1.2)

`fill_parameters` is a deterministic closed-form weight filler (a function of the
parameter name and flat index only) so that golden fixtures need not store weights.
"""
import math
import zlib
from typing import Dict

import numpy as np


def make_cloud(rng: np.random.Generator, n: int, kind: str) -> np.ndarray:
    if kind == "slab":
        pc = np.stack([rng.uniform(-50, 50, n), rng.normal(0, 1, n), rng.uniform(-50, 50, n)])
    elif kind == "cube":
        pc = rng.uniform(-50, 50, (3, n))
    elif kind == "sphere":
        v = rng.normal(0, 1, (3, n))
        pc = v / np.linalg.norm(v, axis=0, keepdims=True) * rng.uniform(0.8, 1.2, (1, n))
    elif kind.startswith("slab:"):          # slab with a custom half-extent, e.g. "slab:8"
        h = float(kind.split(":")[1])
        pc = np.stack([rng.uniform(-h, h, n), rng.normal(0, 1, n), rng.uniform(-h, h, n)])
    else:
        raise ValueError(kind)
    return pc.astype(np.float32)


def make_normals(rng: np.random.Generator, n: int, cs: int) -> np.ndarray:
    """Unit normals (+ curvature U(0,1) for Cs=4, + reflectance U(0,1) for Cs=5)."""
    v = rng.normal(0, 1, (3, n))
    sn = v / np.linalg.norm(v, axis=0, keepdims=True)
    extra = rng.uniform(0, 1, (max(cs - 3, 0), n))
    return np.concatenate([sn, extra], 0)[:cs].astype(np.float32)


def make_pair_batch(seed: int, pairs: int, n: int, m: int, cs: int, kind: str = "slab"
                    ) -> Dict[str, np.ndarray]:
    """One detector batch: `pairs` (src, dst) pairs.  dst is an independent re-sample of the
    same generator transformed like the loader does with rot_type='2d': yaw U(0,2pi) about y,
    scale 1, shift U(-0.5,0.5)^3.  Nodes = first m entries of a seeded permutation of each
    cloud (FPS runs loader-side, outside the path)."""
    rng = np.random.default_rng(seed)
    out = {k: [] for k in ("src_pc", "src_sn", "src_node", "dst_pc", "dst_sn", "dst_node",
                           "R", "scale", "shift")}
    for _ in range(pairs):
        src = make_cloud(rng, n, kind)
        base = make_cloud(rng, n, kind)
        yaw = rng.uniform(0, 2 * math.pi)
        c, s = math.cos(yaw), math.sin(yaw)
        R = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=np.float32)
        shift = rng.uniform(-0.5, 0.5, (3, 1)).astype(np.float32)
        dst = (R @ base + shift).astype(np.float32)
        out["src_pc"].append(src)
        out["dst_pc"].append(dst)
        out["src_sn"].append(make_normals(rng, n, cs))
        out["dst_sn"].append(make_normals(rng, n, cs))
        out["src_node"].append(src[:, rng.permutation(n)[:m]])
        out["dst_node"].append(dst[:, rng.permutation(n)[:m]])
        out["R"].append(R)
        out["scale"].append(np.float32(1.0))
        out["shift"].append(shift)
    return {k: np.ascontiguousarray(np.stack(v), dtype=np.float32) for k, v in out.items()}


def fill_parameters(named_shapes: Dict[str, tuple], head_std: float = 0.05
                    ) -> Dict[str, np.ndarray]:
    """Deterministic non-trivial values for every parameter / BN buffer of a detector:
    conv weights ~ N(0, sqrt(2/fan_in)), conv biases ~ 0.05 N(0,1), BN gamma ~ 1+0.1 N,
    BN beta ~ 0.1 N, running_mean ~ 0.1 N, running_var ~ 1+0.1|N|; the last head layer
    (mlp3) uses std `head_std` so keypoint offsets are visible in fixtures."""
    out = {}
    for name, shape in named_shapes.items():
        rng = np.random.default_rng(zlib.crc32(name.encode()))
        n = rng.normal(0, 1, shape)
        if name.endswith("num_batches_tracked"):
            v = np.zeros(shape, dtype=np.int64)
            out[name] = v
            continue
        if name.endswith("conv.weight"):
            fan_in = int(np.prod(shape[1:]))
            std = head_std if name.startswith("mlp3.") else math.sqrt(2.0 / fan_in)
            v = n * std
        elif name.endswith("conv.bias"):
            v = 0.05 * n
        elif name.endswith("norm.weight"):
            v = 1.0 + 0.1 * n
        elif name.endswith("norm.bias") or name.endswith("running_mean"):
            v = 0.1 * n
        elif name.endswith("running_var"):
            v = 1.0 + 0.1 * np.abs(n)
        else:
            v = 0.1 * n
        out[name] = v.astype(np.float32)
    return out


def grad_projections(name: str, g: np.ndarray, count: int = 4) -> np.ndarray:
    """`count` seeded +-1 (Rademacher) projections of a flat gradient, accumulated in float64 and divided by
    sqrt(len): every entry of the tensor enters every projection, so a handful of numbers pins the whole
    gradient in a fixture.  The sign vectors are a function of the parameter name only."""
    g = np.asarray(g, dtype=np.float64).ravel()
    rng = np.random.default_rng(zlib.crc32(("proj/" + name).encode()))
    out = np.empty(count, dtype=np.float64)
    for i in range(count):
        signs = rng.integers(0, 2, g.size).astype(np.float64) * 2.0 - 1.0
        out[i] = float(signs @ g) / math.sqrt(max(g.size, 1))
    return out
