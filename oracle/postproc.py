"""oracle/postproc.py -- TEST INFRASTRUCTURE ONLY.

numpy restatements of the two steps either side of the detector path, pinned by golden
vectors produced by executing the reference's own functions (tests/golden/make_golden.py extracts them from the
reference files with `ast`, nothing is copied into this repository).
"""
import numpy as np


def fps_indices(pts: np.ndarray, first: int, k: int) -> np.ndarray:
    """FarthestSampler.sample (data/kitti_detector_loader.py:69-83) returning INDICES: pts [n,3] float32,
    start at `first` (the reference draws it with np.random.randint, :78), float64 distances (the reference's
    zeros() buffer is float64, :77), first arg-max, running np.minimum."""
    p = pts.astype(np.float64)
    out = np.zeros(k, dtype=np.int32)
    out[0] = first
    dist = ((p[first] - p) ** 2).sum(axis=1)
    for i in range(1, k):
        j = int(np.argmax(dist))
        out[i] = j
        dist = np.minimum(dist, ((p[j] - p) ** 2).sum(axis=1))
    return out


def nms_order(kp: np.ndarray, sigma: np.ndarray, radius: float) -> np.ndarray:
    """nms() (evaluation/save_keypoints.py:180-216) returning the kept INDICES in pick order: kp [M,3] float32,
    sigma [M] float32; smallest remaining sigma first (first index on ties, np.argmin), survivors are the points
    with float32 np.linalg.norm distance > radius."""
    idx = np.arange(kp.shape[0])
    kp, sigma = kp.copy(), sigma.copy()
    kept = []
    if radius < 0.01:                       # :188-189: no suppression at all
        return idx.astype(np.int32)
    while kp.shape[0] > 0:
        m = int(np.argmin(sigma, axis=0))
        kept.append(int(idx[m]))
        d = np.linalg.norm(kp[m:m + 1, :] - kp, axis=1, keepdims=False)
        mask = d > radius
        kp, sigma, idx = kp[mask], sigma[mask], idx[mask]
    return np.asarray(kept, dtype=np.int32)


def export_keypoints(kp: np.ndarray, sigma: np.ndarray, radius: float, desired_num: int) -> np.ndarray:
    """NMS, then the `desired_num` smallest sigmas (save_keypoints.py:343-351), as the float32 M' x 3 row-major
    array the reference writes with tofile (:392-393)."""
    order = nms_order(kp, sigma, radius)
    k2, s2 = kp[order], sigma[order]
    sel = np.argsort(s2)[:min(desired_num, k2.shape[0])]
    return k2[sel].astype(np.float32)
