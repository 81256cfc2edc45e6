"""Tensor-level wrappers over the C ABI (include/usip_hip.h).

Every function takes contiguous device tensors, allocates its outputs with torch (device
memory + stream plumbing only), enqueues the HIP kernels on torch's CURRENT stream and
returns without synchronising.  Host tensors raise RuntimeError: there is no CPU path here.
"""
import ctypes
import os
from typing import Optional

import torch

from . import _lib, prof


def _stream(t: torch.Tensor):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _need(t: torch.Tensor, name: str, dtype):
    # mirrors CHECK_INPUT of the reference (index_max.cpp:119-121, ball_query.cpp:10-12):
    # device + contiguous, reported as RuntimeError; dtype is checked too (the reference
    # would read garbage).
    if not isinstance(t, torch.Tensor):
        raise RuntimeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("%s must be a CUDA tensor/variable" % name)
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous" % name)
    if t.dtype != dtype:
        raise RuntimeError("%s must be %s (got %s)" % (name, dtype, t.dtype))


def _ptr(t: torch.Tensor):
    return ctypes.c_void_p(t.data_ptr())


def index_max(data: torch.Tensor, index: torch.Tensor, K: int) -> torch.Tensor:
    """a-1: data f32 [B,C,N], index i32 [B,N] -> i32 [B,C,K] (index_max_cuda.cu:83-98)."""
    K = int(K)
    _need(data, "data", torch.float32)
    _need(index, "index", torch.int32)
    if data.dim() != 3 or index.dim() != 2 or index.shape[0] != data.shape[0] or index.shape[1] != data.shape[2]:
        raise RuntimeError("index_max: expected data [B,C,N] and index [B,N]")
    B, C, N = data.shape
    out = torch.empty((B, C, int(K)), dtype=torch.int32, device=data.device)
    with torch.cuda.device(data.device), prof.kernel("index_max", 4.0 * (B * C * N + B * N + B * C * K)):
        _lib.check(_lib.lib().usip_index_max_f32(_ptr(data), _ptr(index), _ptr(out), B, C, N, int(K),
                                                 _stream(data)), "usip_index_max_f32")
    return out


def index_max_values(data: torch.Tensor, index: torch.Tensor, count, K: int, C: Optional[int] = None):
    """index_max plus the gather + mask the reference applies to it (networks.py:117-118): -> (max_idx i32 [B,C,K],
    data[b,c,max_idx] * (count > 0) f32 [B,C,K]).  data [B,Ctot,N]; C: use only the first C channels."""
    K = int(K)
    _need(data, "data", torch.float32)
    _need(index, "index", torch.int32)
    if data.dim() != 3 or index.dim() != 2 or index.shape[0] != data.shape[0] or index.shape[1] != data.shape[2]:
        raise RuntimeError("index_max_values: expected data [B,C,N] and index [B,N]")
    B, Ctot, N = data.shape
    C = Ctot if C is None else int(C)
    if count is not None:
        _need(count, "count", torch.int32)
        if tuple(count.shape) != (B, K):
            raise RuntimeError("index_max_values: count must be [B,K]")
    idx = torch.empty((B, C, K), dtype=torch.int32, device=data.device)
    val = torch.empty((B, C, K), dtype=torch.float32, device=data.device)
    with torch.cuda.device(data.device), prof.kernel("index_max", 4.0 * (B * C * N + B * N + 2 * B * C * K)):
        _lib.check(_lib.lib().usip_index_max_values_f32(_ptr(data), _ptr(index), _opt(count), _ptr(idx), _ptr(val),
                                                        B, C, Ctot, N, K, _stream(data)), "usip_index_max_values_f32")
    return idx, val


def index_max_values_backward_add_(ddata, g, max_idx, count, coff: int = 0):
    """ddata[b, coff+c, max_idx[b,c,k]] += g[b,c,k] (populated nodes), in place; ddata [B,Ctot,N]."""
    _need(ddata, "ddata", torch.float32)
    _need(g, "g", torch.float32)
    B, Ctot, N = ddata.shape
    _, C, K = g.shape
    with torch.cuda.device(g.device), prof.kernel("index_max_bwd", 16.0 * B * C * K):
        _lib.check(_lib.lib().usip_index_max_values_backward_add_f32(_ptr(g), _ptr(max_idx), _opt(count), _ptr(ddata),
                                                                     B, C, Ctot, int(coff), N, K, _stream(g)),
                   "usip_index_max_values_backward_add_f32")
    return ddata


def index_max_values_backward(g, max_idx, count, index, N: int, src=None, soff: int = 0):
    """dz [B,C,N] = (src[:, soff:soff+C] if src is given else 0) + scatter of g [B,C,K] to max_idx (populated nodes):
    the gradient of index_max_values' masked values as one dense pass."""
    _need(g, "g", torch.float32)
    B, C, K = g.shape
    dz = torch.empty((B, C, int(N)), dtype=torch.float32, device=g.device)
    if src is not None:
        _need(src, "src", torch.float32)
    with torch.cuda.device(g.device), prof.kernel("index_max_bwd", 4.0 * B * C * N * (2 if src is not None else 1) + 4.0 * B * N):
        _lib.check(_lib.lib().usip_index_max_values_backward_f32(_ptr(g), _ptr(max_idx), _opt(count), _ptr(index),
                                                                 _opt(src), src.shape[1] if src is not None else 0,
                                                                 int(soff), _ptr(dz), B, C, int(N), K, _stream(g)),
                   "usip_index_max_values_backward_f32")
    return dz


def csr_by_index(idx32: torch.Tensor, N: int):
    """idx i32 [B,P] (values in [0,N)) -> (start i32 [B,N+1], perm i32 [B,P]): the positions sorted by destination."""
    _need(idx32, "idx", torch.int32)
    B, P = idx32.shape
    start = torch.empty((B, int(N) + 1), dtype=torch.int32, device=idx32.device)
    perm = torch.empty((B, P), dtype=torch.int32, device=idx32.device)
    with torch.cuda.device(idx32.device), prof.kernel("csr_by_index", 12.0 * B * P):
        _lib.check(_lib.lib().usip_csr_by_index_i32(_ptr(idx32), _ptr(start), _ptr(perm), B, P, int(N),
                                                    _stream(idx32)), "usip_csr_by_index_i32")
    return start, perm


def segment_sum_supported(N: int, P: int) -> bool:
    return bool(_lib.lib().usip_segment_sum_supported(int(N), int(P))) and int(N) <= 1820


def segment_sum(src, start, perm, C: int, coff: int = 0):
    """dx[b,c,n] = sum over segment n of src[b, coff+c, perm[b,j]]; src [B,Ctot,P] (any trailing shape) -> [B,C,N]."""
    _need(src, "src", torch.float32)
    B, Ctot = src.shape[0], src.shape[1]
    P = perm.shape[1]
    N = start.shape[1] - 1
    dx = torch.empty((B, int(C), N), dtype=torch.float32, device=src.device)
    with torch.cuda.device(src.device), prof.kernel("segment_sum", 4.0 * B * (C * (P + N) + P)):
        _lib.check(_lib.lib().usip_segment_sum_f32(_ptr(src), _ptr(start), _ptr(perm), _ptr(dx), B, int(C), N, P,
                                                   Ctot, int(coff), _stream(src)), "usip_segment_sum_f32")
    return dx


def index_max_geometry(B: int, C: int, N: int, K: int):
    """(channel rows per workgroup, prefetch depth, threads) usip_index_max_f32 picks for this shape
    (csrc/index_max.hip); lets a profiler name the launch."""
    rows = B * C
    ch = 2 if (C % 2 == 0 and rows // 2 >= 512 and 2 * K * 8 <= 65536) else 1
    u = 2 if N >= 4096 else 1
    t = 256
    tch, tu, tt = (_lib.lib().usip_tuning_value(i) for i in (0, 1, 5))
    if tch > 0 and C % tch == 0 and tch * K * 8 <= 65536:
        ch = tch
    if tu > 0:
        u = tu
    if tt in (512, 1024):
        t = tt
    if t > 256:
        u = 2 if u >= 2 else 1
    else:
        u = 4 if (u >= 4 and ch <= 4) else (2 if u >= 2 else 1)
    return ch, u, t


def ball_query(dist: torch.Tensor, radius: float, K: int) -> torch.Tensor:
    """a-2: dist f32 [B,M,N] -> i32 [B,M,K] (ball_query_cuda.cu:53-70)."""
    K = int(K)
    _need(dist, "node_to_point_dist", torch.float32)
    if dist.dim() != 3:
        raise RuntimeError("ball_query: expected dist [B,M,N]")
    B, M, N = dist.shape
    out = torch.empty((B, M, int(K)), dtype=torch.int32, device=dist.device)
    with torch.cuda.device(dist.device), prof.kernel("ball_query", 4.0 * (B * M * N + B * M * K)):
        _lib.check(_lib.lib().usip_ball_query_f32(_ptr(dist), _ptr(out), float(radius), int(K), B, M, N,
                                                  _stream(dist)), "usip_ball_query_f32")
    return out


def pairwise_dist(a: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """|a[b,:,m] - x[b,:,n]| -> f32 [B,M,N]; a [B,3,M], x [B,3,N]."""
    _need(a, "a", torch.float32)
    _need(x, "x", torch.float32)
    if a.dim() != 3 or x.dim() != 3 or a.shape[1] != 3 or x.shape[1] != 3 or a.shape[0] != x.shape[0]:
        raise RuntimeError("pairwise_dist: expected a [B,3,M], x [B,3,N]")
    B, _, M = a.shape
    N = x.shape[2]
    out = torch.empty((B, M, N), dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device), prof.kernel("pairwise_dist", 4.0 * (B * M * N + 3 * B * (M + N))):
        _lib.check(_lib.lib().usip_pairwise_dist_f32(_ptr(a), _ptr(x), _ptr(out), B, M, N, _stream(a)),
                   "usip_pairwise_dist_f32")
    return out


def ball_query_coords(node: torch.Tensor, x: torch.Tensor, radius: float, K: int) -> torch.Tensor:
    """f-2: fused pairwise_dist + ball_query; node [B,3,M], x [B,3,N] -> i32 [B,M,K]."""
    K = int(K)
    _need(node, "node", torch.float32)
    _need(x, "x", torch.float32)
    if node.dim() != 3 or x.dim() != 3 or node.shape[1] != 3 or x.shape[1] != 3 or node.shape[0] != x.shape[0]:
        raise RuntimeError("ball_query_coords: expected node [B,3,M], x [B,3,N]")
    B, _, M = node.shape
    N = x.shape[2]
    out = torch.empty((B, M, int(K)), dtype=torch.int32, device=node.device)
    with torch.cuda.device(node.device), prof.kernel("ball_query_coords", 4.0 * (3 * B * (M + N) + B * M * K),
                                                    8.0 * B * M * N):
        _lib.check(_lib.lib().usip_ball_query_coords_f32(_ptr(node), _ptr(x), _ptr(out), float(radius), int(K),
                                                         B, M, N, _stream(node)), "usip_ball_query_coords_f32")
    return out


def _need_pts(t, name):
    _need(t, name, torch.float32)
    if t.dim() != 3 or t.shape[1] != 3:
        raise RuntimeError("%s must be [B,3,N]" % name)


def som_assign(x: torch.Tensor, node: torch.Tensor) -> torch.Tensor:
    """a-3: nearest SOM node of every point. x [B,3,N], node [B,3,M] -> min_idx i32 [B,N]."""
    _need_pts(x, "x")
    _need_pts(node, "node")
    B, _, N = x.shape
    M = node.shape[2]
    out = torch.empty((B, N), dtype=torch.int32, device=x.device)
    with torch.cuda.device(x.device), prof.kernel("som_assign", 4.0 * (3 * B * (M + N) + B * N), 9.0 * B * M * N):
        _lib.check(_lib.lib().usip_som_assign_f32(_ptr(x), _ptr(node), _ptr(out), B, N, M, _stream(x)),
                   "usip_som_assign_f32")
    return out


def som_cluster(x: torch.Tensor, min_idx: torch.Tensor, M: int, decenter: bool = True, csr=None):
    """a-4: (cluster_mean f32 [B,3,M], count i32 [B,M], x_decentered f32 [B,3,N] or None).  csr = (start, perm) of
    min_idx from csr_by_index: the cluster sums walk the sorted segments instead of scanning all assignments."""
    _need_pts(x, "x")
    _need(min_idx, "min_idx", torch.int32)
    B, _, N = x.shape
    if tuple(min_idx.shape) != (B, N):
        raise RuntimeError("som_cluster: min_idx must be [B,N] = %s (got %s)" % ((B, N), tuple(min_idx.shape)))
    mean = torch.empty((B, 3, int(M)), dtype=torch.float32, device=x.device)
    count = torch.empty((B, int(M)), dtype=torch.int32, device=x.device)
    dec = torch.empty_like(x) if decenter else None
    if csr is not None:
        with torch.cuda.device(x.device), prof.kernel("som_cluster", 4.0 * (8 * B * N + 5 * B * M)):
            _lib.check(_lib.lib().usip_som_cluster_csr_f32(_ptr(x), _ptr(min_idx), _ptr(csr[0]), _ptr(csr[1]), _ptr(mean),
                                                           _ptr(count), _ptr(dec) if decenter else None, B, N, int(M),
                                                           _stream(x)), "usip_som_cluster_csr_f32")
        return mean, count, dec
    with torch.cuda.device(x.device), prof.kernel("som_cluster", 4.0 * (7 * B * N + 4 * B * M)):
        _lib.check(_lib.lib().usip_som_cluster_f32(_ptr(x), _ptr(min_idx), _ptr(mean), _ptr(count),
                                                   _ptr(dec) if decenter else None, B, N, int(M), _stream(x)),
                   "usip_som_cluster_f32")
    return mean, count, dec


def nearest(a: torch.Tensor, b: torch.Tensor):
    """a-9/a-10 core: (min_j |a_i - b_j| f32 [B,Ma], first arg-min i32 [B,Ma])."""
    _need_pts(a, "a")
    _need_pts(b, "b")
    B, _, Ma = a.shape
    Nb = b.shape[2]
    d = torch.empty((B, Ma), dtype=torch.float32, device=a.device)
    arg = torch.empty((B, Ma), dtype=torch.int32, device=a.device)
    ws = int(_lib.lib().usip_nearest_workspace(B, Ma, Nb))
    ws_d = torch.empty(ws, dtype=torch.float32, device=a.device) if ws else None
    ws_j = torch.empty(ws, dtype=torch.int32, device=a.device) if ws else None
    with torch.cuda.device(a.device), prof.kernel("nearest", 4.0 * (3 * B * (Ma + Nb) + 2 * B * Ma), 8.0 * B * Ma * Nb):
        _lib.check(_lib.lib().usip_nearest_f32(_ptr(a), _ptr(b), _ptr(d), _ptr(arg), _opt(ws_d), _opt(ws_j),
                                               B, Ma, Nb, _stream(a)), "usip_nearest_f32")
    return d, arg


def chamfer_prob(a, J, c, I, sigma_src, sigma_dst):
    """(loss, chamfer_pure, chamfer_weighted) as a 3-vector from the nearest-neighbour minima a [B,M] / J i32
    and c [B,N] / I i32 and the two sigma maps (models/losses.py:82-99)."""
    for t, n in ((a, "a"), (c, "c"), (sigma_src, "sigma_src"), (sigma_dst, "sigma_dst")):
        _need(t, n, torch.float32)
    _need(J, "J", torch.int32)
    _need(I, "I", torch.int32)
    B, M = a.shape
    N = c.shape[1]
    if J.shape != a.shape or I.shape != c.shape or sigma_src.shape != a.shape or sigma_dst.shape != c.shape:
        raise RuntimeError("chamfer_prob: shapes do not match")
    out = torch.empty(3, dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device), prof.kernel("chamfer_prob", 4.0 * 3 * B * (M + N), 0.0):
        _lib.check(_lib.lib().usip_chamfer_prob_f32(_ptr(a), _ptr(J), _ptr(c), _ptr(I), _ptr(sigma_src),
                                                    _ptr(sigma_dst), _ptr(out), B, M, N, _stream(a)),
                   "usip_chamfer_prob_f32")
    return out


def chamfer_prob_backward(gloss, a, J, c, I, sigma_src, sigma_dst):
    """-> (da, dc, dsigma_src, dsigma_dst) for the upstream gradient `gloss` (0-dim device tensor) of the loss."""
    B, M = a.shape
    N = c.shape[1]
    _need(gloss, "gloss", torch.float32)
    da, dc = torch.empty_like(a), torch.empty_like(c)
    dss, dsd = torch.empty_like(sigma_src), torch.empty_like(sigma_dst)
    with torch.cuda.device(a.device), prof.kernel("chamfer_prob_bwd", 4.0 * 7 * B * (M + N), 0.0):
        _lib.check(_lib.lib().usip_chamfer_prob_backward_f32(_ptr(gloss), _ptr(a), _ptr(J), _ptr(c), _ptr(I),
                                                             _ptr(sigma_src), _ptr(sigma_dst), _ptr(da), _ptr(dc),
                                                             _ptr(dss), _ptr(dsd), B, M, N, _stream(a)),
                   "usip_chamfer_prob_backward_f32")
    return da, dc, dss, dsd


def bn_group_dy_sum(gsum, coef4, K: int):
    """sum over each neighbourhood's K positions of dY, [nb,C,G], from gsum = (sum dYhat, sum y) (each [nb,C,G])."""
    g0, g1 = gsum[0], gsum[1]
    nb, C, G = g0.shape
    out = torch.empty_like(g0)
    with torch.cuda.device(g0.device), prof.kernel("bn_group_dy_sum", 12.0 * g0.numel()):
        _lib.check(_lib.lib().usip_bn_group_dy_sum_f32(_ptr(g0), _ptr(g1), _ptr(coef4), _ptr(out), nb, C, G, int(K),
                                                       _stream(g0)), "usip_bn_group_dy_sum_f32")
    return out


def detector_head(ks, centre, sigma_lower_bound: float):
    """ks [B,4,M] (mlp3 output), centre [B,3,M] -> (keypoints [B,3,M], sigmas [B,M]) (networks.py:150-154)."""
    _need(ks, "ks", torch.float32)
    _need(centre, "centre", torch.float32)
    B, _, M = ks.shape
    kp = torch.empty((B, 3, M), dtype=torch.float32, device=ks.device)
    sg = torch.empty((B, M), dtype=torch.float32, device=ks.device)
    with torch.cuda.device(ks.device), prof.kernel("head", 4.0 * 11 * B * M):
        _lib.check(_lib.lib().usip_detector_head_f32(_ptr(ks), _ptr(centre), float(sigma_lower_bound), _ptr(kp), _ptr(sg),
                                                     B, M, _stream(ks)), "usip_detector_head_f32")
    return kp, sg


def detector_head_backward(g_kp, g_sg, ks):
    B, _, M = ks.shape
    g = torch.empty_like(ks)
    with torch.cuda.device(ks.device), prof.kernel("head_bwd", 4.0 * 9 * B * M):
        _lib.check(_lib.lib().usip_detector_head_backward_f32(_opt(g_kp), _opt(g_sg), _ptr(ks), _ptr(g), B, M,
                                                              _stream(ks)), "usip_detector_head_backward_f32")
    return g


def rigid_transform(x, R, scale, shift, transpose: bool = False):
    """(R*scale) . x + shift per cloud (keypoint_detector.py:182-184); transpose: (R*scale)^T . x (the backward)."""
    _need(x, "x", torch.float32)
    B, _, M = x.shape
    out = torch.empty_like(x)
    with torch.cuda.device(x.device), prof.kernel("rigid_transform", 4.0 * 6 * B * M):
        _lib.check(_lib.lib().usip_rigid_transform_f32(_ptr(x), _ptr(R), _ptr(scale), _opt(shift), _ptr(out),
                                                       int(bool(transpose)), B, M, _stream(x)),
                   "usip_rigid_transform_f32")
    return out


def detector_loss_combine(d, chamfer, alpha: float):
    """d [2B,M] keypoint-to-cloud distances (src rows first), chamfer: 0-dim/1-element loss tensor ->
    3-vector (loss, alpha*mean(d_src), alpha*mean(d_dst)) (keypoint_detector.py:196-204)."""
    _need(d, "d", torch.float32)
    out = torch.empty(3, dtype=torch.float32, device=d.device)
    with torch.cuda.device(d.device), prof.kernel("loss_combine", 4.0 * d.numel()):
        _lib.check(_lib.lib().usip_detector_loss_combine_f32(_ptr(d), _ptr(chamfer), float(alpha), _ptr(out),
                                                             d.numel() // 2, _stream(d)),
                   "usip_detector_loss_combine_f32")
    return out


def fill_scaled(g, factor: float, shape):
    """A tensor of `shape` filled with g[0] * factor (g: device scalar)."""
    out = torch.empty(shape, dtype=torch.float32, device=g.device)
    with torch.cuda.device(g.device), prof.kernel("fill_scaled", 4.0 * out.numel()):
        _lib.check(_lib.lib().usip_fill_scaled_f32(_ptr(g), float(factor), _ptr(out), out.numel(), _stream(g)),
                   "usip_fill_scaled_f32")
    return out


# --------------------------------------------------------------------------- shared MLP
def _opt(t):
    return _ptr(t) if t is not None else None


# "f32": fp32 MFMA (the parity mode and the headline).  "bf16": operands rounded to bf16 inside the GEMM and
# weight-gradient kernels, fp32 accumulate, everything else fp32 (BASELINE.json configs[1] perf mode).
#   "f32x3": fp32-accurate products on the bf16 matrix cores (three bf16 planes per operand, six plane products) for
#   the matrix-bound launches, the fp32 kernel for the rest; results equal "f32" to fp32 rounding.
#   "f32x2": as f32x3, and the launches whose streamed operand has a rigorous bound (BatchNorm outputs forward, the
#   BatchNorm-backward operand with the bound usip_bn_backward_reduce_f32 computes) use TWO fp16 planes and THREE plane
#   products instead (half the matrix work, error at the same level); the others run the f32x3 kernel.
MATMUL_MODES = ("f32", "bf16", "f32x3", "f32x2")


def _x3_family() -> bool:
    return _matmul_mode in ("f32x3", "f32x2")
_matmul_mode = "f32"
import os as _os                                             # noqa: E402
if _os.environ.get("USIP_MATMUL_MODE"):                      # lets the whole GPU test suite run under another mode
    if _os.environ["USIP_MATMUL_MODE"] not in MATMUL_MODES:
        raise ValueError("USIP_MATMUL_MODE must be one of %s" % (MATMUL_MODES,))
    _matmul_mode = _os.environ["USIP_MATMUL_MODE"]


def set_matmul_mode(mode: str) -> str:
    """Select the multiply precision of the shared-MLP kernels; returns the previous mode."""
    global _matmul_mode
    if mode not in MATMUL_MODES:
        raise ValueError("matmul mode must be one of %s" % (MATMUL_MODES,))
    prev, _matmul_mode = _matmul_mode, mode
    return prev


def matmul_mode() -> str:
    return _matmul_mode


# f32x3 mode: split images of the weight operands (usip_mlp_split3_f32), keyed by (storage pointer, lda, column
# offset, M, K).  The training step sets this to a dict for the duration of one forward + backward (weights do not
# change in between), so every weight is split once per step; None: split at every call.
PLANES_CACHE = None


def weight_planes(At: torch.Tensor, a_offset: int, M: int, K: int, npl: int = 3, P: int = 0, nb: int = 1) -> torch.Tensor:
    """Split image (uint8 tensor) of the M x K operand A[m][k] = At[k, a_offset + m]: three bf16 planes for
    usip_mlp_gemm_x3p_f32 (npl 3) or two fp16 planes + scale for usip_mlp_gemm_x2h_f32 (npl 2), tiled for a launch over
    nb clouds of P positions (usip_mlp_x3p_tile_rows: small launches take 128-row tiles)."""
    lda = At.shape[1]
    rows = int(_lib.lib().usip_mlp_x3p_tile_rows(int(M), int(P), int(nb)))
    key = (At.data_ptr(), lda, int(a_offset), int(M), int(K), int(npl), rows)
    if PLANES_CACHE is not None and key in PLANES_CACHE:
        return PLANES_CACHE[key][1]
    nbytes = int(_lib.lib().usip_mlp_split3_bytes(M, K))
    planes = torch.empty(nbytes, dtype=torch.uint8, device=At.device)
    fn = "usip_mlp_split2h_f32" if npl == 2 else "usip_mlp_split3_f32"
    with torch.cuda.device(At.device), prof.kernel("weight_split3", 4.0 * M * K + nbytes):
        _lib.check(getattr(_lib.lib(), fn)(ctypes.c_void_p(At.data_ptr() + 4 * int(a_offset)), lda, M, K, rows,
                                           _ptr(planes), _stream(At)), fn)
    if PLANES_CACHE is not None:
        PLANES_CACHE[key] = (At, planes)      # holding At keeps its storage (the key) from being reused meanwhile
    return planes


class PlanesPlan:
    """The weight operands a step asks weight_planes() for, with persistent split images, refreshed by ONE launch at
    the start of the step (twelve 5-us launches otherwise).  Built from the PLANES_CACHE of a completed step; only
    operands inside the given persistent storages (parameters, their K-major copies) qualify -- their addresses are
    the cache keys, and a temporary's address could be handed out again."""

    def __init__(self, cache, storages):
        import numpy as np
        spans = [(t.data_ptr(), t.data_ptr() + t.numel() * t.element_size()) for t in storages if t is not None]
        self.entries, rows, blocks = {}, [], 0
        for key, (At, planes) in cache.items():
            ptr, lda, off, M, K, npl, trows = key
            if not any(lo <= ptr < hi for lo, hi in spans):
                continue
            self.entries[key] = (At, planes)
            rows.append((ptr + 4 * off, planes.data_ptr(), lda, M, K, blocks, trows, 2 if npl == 2 else 0))
            blocks += int(_lib.lib().usip_mlp_split3_blocks(M, K, trows))
        self.blocks = blocks
        dt = np.dtype([("At", "<u8"), ("planes", "<u8"), ("lda", "<i4"), ("M", "<i4"), ("K", "<i4"), ("first", "<i4"),
                       ("tile_rows", "<i4"), ("reserved", "<i4")])
        table = np.array(rows, dtype=dt) if rows else np.zeros(0, dtype=dt)
        dev = next(iter(self.entries.values()))[1].device if rows else None
        self.table = torch.from_numpy(table.view(np.uint8).copy()).to(dev) if rows else None

    def refresh(self):
        """Split every operand of the plan now; -> a PLANES_CACHE holding the fresh images."""
        if self.table is not None:
            with torch.cuda.device(self.table.device), prof.kernel("weight_split3", 0.0):
                _lib.check(_lib.lib().usip_mlp_split3_multi_f32(_ptr(self.table), len(self.entries), self.blocks,
                                                                _stream(self.table)), "usip_mlp_split3_multi_f32")
        return dict(self.entries)


# The narrow forward layers run the streaming kernel; USIP_NARROW_FWD=0 sends them through the generic tile kernel
# again (A/B measurement, tools/ab_env.sh).
NARROW_FWD = _os.environ.get("USIP_NARROW_FWD", "1") not in ("0", "off")
# f32x2 mode: the 128-wide layers on the register-resident-weights kernel; USIP_X2R=0 for A/B runs
X2R = _os.environ.get("USIP_X2R", "1") not in ("0", "off")
X2R_NARROW = _os.environ.get("USIP_X2R_NARROW", "1") not in ("0", "off")


def mlp_gemm(At: torch.Tensor, X: torch.Tensor, bias=None, want_stats: bool = False, pro: int = 0,
             X2=None, coef=None, tag: str = "fwd", rowbias=None, rb_group: int = 1,
             M: Optional[int] = None, a_offset: int = 0, pool=None, a_trans: bool = False, out=None,
             out_row_offset: int = 0, red=None, red_group: int = 0):
    """Y[b] = At^T . pro(X[b]) + bias (+ rowbias[b][:, p // rb_group]).
    At: K-major matrix operand, [K, lda] storage; the GEMM uses columns [a_offset, a_offset + M)
    (default: all of them).  X [nb,K,P] -> Y [nb,M,P] (+ stats [2,M,tiles] when want_stats).
    red = (y_prev [nb,M,P], coef_prev [>=4,M]) (data-gradient launches, pro 2 / 3): Y is the gradient of the lazily
    activated output of the layer with that pre-BN tensor and those coefficients; when the launch takes the direct f32x2
    kernel the call returns a THIRD value, the RedSums its epilogue left for that layer's BatchNorm backward (with .gsum
    [2,nb,M,P/red_group] when red_group is 16 or 32), else None as the third value."""
    _need(At, "At", torch.float32)
    if a_trans:                               # At is [M_total, lda]: the operand is rows [m0, m0+M) x cols [a_offset, a_offset+K)
        M_rows, lda = At.shape
        M = M_rows if M is None else int(M)
        K = X2.shape[1] if X is None else X.shape[1]
    else:
        K, lda = At.shape
        M = lda if M is None else int(M)
    pool_dp = pool_arg = None
    pool_group = 0
    if pool is not None:                      # pro == 3: (dpooled [nb,K,G], arg i32 [nb,K,G], group); X unused
        pool_dp, pool_arg, pool_group = pool
        X = X2
    _need(X, "X", torch.float32)
    nb, Kx, P = X.shape
    if Kx != K or a_offset < 0 or a_offset + (K if a_trans else M) > lda:
        raise RuntimeError("mlp_gemm: operand shapes do not match (At %s, X %s, M %d, offset %d)"
                           % (tuple(At.shape), tuple(X.shape), M, a_offset))
    y_rows = 0
    if out is None:
        Y = torch.empty((nb, M, P), dtype=torch.float32, device=X.device)
        y_ptr = _ptr(Y)
    else:                                     # rows [out_row_offset, out_row_offset + M) of a wider [nb, rows, P] tensor
        _need(out, "out", torch.float32)
        if out.dim() != 3 or out.shape[0] != nb or out.shape[2] != P or out_row_offset < 0 \
                or out_row_offset + M > out.shape[1]:
            raise RuntimeError("mlp_gemm: out %s does not hold rows [%d, %d) of [%d, *, %d]"
                               % (tuple(out.shape), out_row_offset, out_row_offset + M, nb, P))
        Y, y_rows = out, out.shape[1]
        y_ptr = ctypes.c_void_p(out.data_ptr() + 4 * int(out_row_offset) * P)
    a_ptr = ctypes.c_void_p(At.data_ptr() + 4 * int(a_offset))
    # narrow layers (64 inputs, 64 / 128 outputs) at many positions: the streaming kernel (csrc/narrow_fwd.hip)
    # f32x2 mode: the 64 -> 128 layer (the feature half of conv4, with its row bias) goes to the register-resident f32x2
    # kernel -- the streaming kernel's fp32 MFMAs make it matrix-bound there (stand-alone 79-146 us depending on the box
    # against 82-109; in the step the same or better).  USIP_X2R_NARROW=0 for A/B runs.
    rb_ok = rowbias is None or (rb_group % 32 == 0 and P % rb_group == 0)
    x2r_direct = (X2R and X2R_NARROW and _matmul_mode == "f32x2" and pro == 1 and coef is not None and coef.shape[0] >= 4
                  and M == 128 and K == 64 and not a_trans and pool is None and X2 is None and rb_ok
                  and nb * P >= 65536 and K * P < 2 ** 30)
    nf_blocks = 0
    if (NARROW_FWD and not x2r_direct and _matmul_mode != "bf16" and not a_trans and pool is None and X2 is None and pro in (0, 1)
            and K == 64 and M in (64, 128) and X.data_ptr() % 16 == 0
            and (rowbias is None or (rb_group % 32 == 0 and P % rb_group == 0))):
        nf_blocks = int(_lib.lib().usip_mlp_narrow_forward_blocks(M, K, P, nb))
    if nf_blocks:
        stats = torch.empty((2, M, nf_blocks), dtype=torch.float32, device=X.device) if want_stats else None
        with torch.cuda.device(X.device), prof.kernel(
                "shared_mlp_gemm_%s %dx%d" % (tag, M, K), 4.0 * nb * P * (K + M), 2.0 * M * K * nb * P,
                rocprof_key="narrow_fwd_kernel<%d, %s, %s, %s> |wg=%d" % (
                    M // 32, "true" if pro else "false", "true" if want_stats else "false",
                    "true" if rowbias is not None else "false", nf_blocks)):
            _lib.check(_lib.lib().usip_mlp_narrow_forward_f32(a_ptr, lda, _ptr(X), _opt(coef), int(pro), _opt(bias),
                                                              _opt(rowbias), int(rb_group), y_ptr, int(y_rows),
                                                              _opt(stats), M, K, P, nb, _stream(X)),
                       "usip_mlp_narrow_forward_f32")
        return (Y, stats, None) if red is not None else (Y, stats)
    bf16 = _matmul_mode == "bf16"
    fn_name = {"bf16": "usip_mlp_gemm_bf16", "f32x3": "usip_mlp_gemm_f32x3",
               "f32x2": "usip_mlp_gemm_f32x3"}.get(_matmul_mode, "usip_mlp_gemm_f32")
    x3 = _x3_family() and (x2r_direct or bool(_lib.lib().usip_mlp_gemm_f32x3_used(M, K, P, nb)))
    x3p = x3 and not a_trans and K <= (512 if pro >= 2 else 640)       # weight operand split ahead of time
    # two fp16 planes: only where the streamed operand's bound is known (see usip_mlp_gemm_x2h_f32)
    x2h = (x3p and _matmul_mode == "f32x2" and coef is not None and
           ((pro == 1 and coef.shape[0] >= 4 and bound_covers(coef, nb * P)) or (pro >= 2 and coef.shape[0] >= 5)))
    # 128-wide layers: weight fragments resident in registers, persistent workgroups (usip_mlp_gemm_x2r_f32)
    x2r = (x2h and M <= 128 and K <= 128 and X2R and K * P < 2 ** 30          # (usip_mlp_gemm_x2r_f32's own limits)
           and (rowbias is None or (pro == 1 and rb_ok and rb_group >= 32)))
    stats = None
    if want_stats:
        tiles = _lib.lib().usip_mlp_gemm_x2r_tiles(P, nb) if x2r else _lib.lib().usip_mlp_gemm_tiles(M, P, nb)
        stats = torch.empty((2, M, tiles), dtype=torch.float32, device=X.device)

    def _key():
        wm, wn = (1, 4) if M <= 64 else (2, 2)
        tiles = _lib.lib().usip_mlp_gemm_tiles(M, P, nb)
        e = 1 if want_stats else 0
        if bf16:
            return "gemm_bf16_kernel<%d, %d, 16, %d, %d, 1> |wg=%d" % (wm, wn, pro, e, tiles * ((M + wm * 64 - 1) // (wm * 64)))
        if x3p:
            bm = _lib.lib().usip_mlp_x3p_tile_rows(M, P, nb)
            bn = _lib.lib().usip_mlp_x3p_tile_cols(M, P, nb, int(pro), e)
            if x2r:
                return "gemm_x2r_kernel<%d, %d, %s> |wg=%d" % (pro, e, "true" if rowbias is not None else "false",
                                                               min(512, nb * ((P + 63) // 64)))
            if x2h:
                # (tile_cols == 256 -- a measurement knob -- sends the launch to launch_x3p<4, 4, 2> in the C dispatcher
                # before the direct kernel is considered: usip_mlp_gemm_x2h_f32; mirrored here)
                wide = bm == 256 and bn == 256
                if not wide:
                    bn = 128
                if bm == 256 and not wide and (_lib.lib().usip_tuning_value(6) & 15) != 1 and K * P * 4 < 2 ** 31:
                    # csrc/gemm_x2d.hip: <pro, stats, K % 16 != 0, stages of operand loads in flight>; persistent, two
                    # workgroups per CU once there are more tiles than that
                    if red is None and _lib.lib().usip_mlp_gemm_x2f_used(
                            M, K, P, nb, int(pro), e, 1 if bias is not None else 0, int(rb_group) if rowbias is not None else 0,
                            int(pool_group), int(y_rows)):
                        # round 6, csrc/gemm_x2f.hip: <pro, stats, DIRECT>; one persistent workgroup per CU
                        tiles_ = nb * (P // 256) * (M // 256)
                        return "gemm_x2f_kernel<%d, %d, %s> |wg=%d" % (pro, e, "true" if pro >= 2 else "false",
                                                                       tiles_ if (tiles_ <= 256 or tiles_ % 8) else 256)
                    tail = K % 16 != 0
                    depth = 2 if (not tail and (K // 16) % 2 == 0 and (_lib.lib().usip_tuning_value(6) & 15) != 2) else 1
                    tiles_ = nb * ((P + 127) // 128) * ((M + 255) // 256)
                    wg = tiles_ if (tiles_ <= 512 or tiles_ % 8) else 512
                    # (+ the epilogue instantiations: <.., RED, general, DIRECT>; the general one runs DEPTH 1; DIRECT =
                    # data gradients that need nothing but the scale: stores straight from the registers)
                    gen = M % 256 != 0 or P % 128 != 0
                    direct = (not gen and not tail and depth == 2 and e == 0 and red is None and bias is None
                              and rowbias is None and pro >= 2 and (_lib.lib().usip_tuning_value(6) & 15) != 8)
                    return "gemm_x2d_kernel<%d, %d, %s, %d, %s, %s, %s> |wg=%d" % (
                        pro, e, "true" if tail else "false", 1 if gen else depth,
                        "true" if red is not None else "false", "true" if gen else "false",
                        "true" if direct else "false", wg)
            slots = 3 if (bm == 128 and not (_lib.lib().usip_tuning_value(7) & 32)) else 2    # round 5: weight-ring slots
            return "gemm_x3p_kernel<%d, %d, %d, %d, %d, %d> |wg=%d" % (pro, e, bm // 64, bn // 64, 2 if x2h else 3, slots,
                                                                       nb * ((P + bn - 1) // bn) * ((M + bm - 1) // bm))
        if x3:
            return "gemm_bf16_kernel<2, 2, 16, %d, %d, 3> |wg=%d" % (pro, e, nb * ((P + 127) // 128) * ((M + 127) // 128))
        # csrc/shared_mlp.hip mlp_gemm_impl: 32 rows per wave when 128-row tiles would not fill the chip
        tm = 1 if ((M > 64 and nb * ((P + 127) // 128) * ((M + 127) // 128) < 512)
                   or (M <= 32 and nb * ((P + 255) // 256) < 512)) else 2
        bm = wm * 32 * tm
        return "gemm_kernel<%d, %d, 16, %d, %d, %s, %d> |wg=%d" % (wm, wn, pro, e, "true" if P % 4 == 0 else "false", tm,
                                                                   tiles * ((M + bm - 1) // bm))

    planes = weight_planes(At, a_offset, M, K, 2 if x2h else 3, P, nb) if x3p else None
    moved = 0.0
    if x3p:      # bytes into the CUs: per (tile, 16-k stage) the weight planes (L2) + the streamed operand, + the output
        bm_ = int(_lib.lib().usip_mlp_x3p_tile_rows(M, P, nb))
        bn_ = 128 if x2h else int(_lib.lib().usip_mlp_x3p_tile_cols(M, P, nb, int(pro), 1 if want_stats else 0))
        ntiles = nb * ((P + bn_ - 1) // bn_) * ((M + bm_ - 1) // bm_)
        moved = ntiles * ((K + 15) // 16) * ((2 if x2h else 3) * bm_ * 32 + 16 * bn_ * 4 * (2 if pro == 2 else 1)) \
            + 4.0 * nb * M * P
        if x2r:                                               # the weight fragments stay in registers
            moved = 4.0 * nb * P * (K * (2 if pro == 2 else 1) + M)
    with torch.cuda.device(X.device), prof.kernel("shared_mlp_gemm_%s %dx%d" % (tag, M, K),
                                                  4.0 * nb * P * (K * (2 if pro == 2 else 1) + M),
                                                  2.0 * M * K * nb * P, rocprof_key=_key, moved=moved):
        if x2r:
            _lib.check(_lib.lib().usip_mlp_gemm_x2r_f32(_ptr(planes), None if pool is not None else _ptr(X), _opt(X2),
                                                        _opt(coef), int(pro), _opt(bias), _opt(rowbias), int(rb_group),
                                                        _opt(pool_dp), _opt(pool_arg), int(pool_group), y_ptr,
                                                        int(y_rows), _opt(stats), M, K, P, nb, _stream(X)),
                       "usip_mlp_gemm_x2r_f32")
            return (Y, stats, None) if red is not None else (Y, stats)
        red_tiles = 0
        if (red is not None and x2h and not x2r and pro >= 2 and out is None and bias is None and rowbias is None
                and red[1] is not None and red[1].shape[0] >= 4 and tuple(red[0].shape) == (nb, M, P)
                and red[0].is_contiguous() and red[0].data_ptr() % 16 == 0):
            red_tiles = int(_lib.lib().usip_mlp_gemm_x2d_red_tiles(M, K, P, nb, int(red_group)))
        if red_tiles:
            flat = torch.empty(2 * red_tiles * M + red_tiles * (M // 256), dtype=torch.float32, device=X.device)
            gsum = torch.empty((2, nb, M, P // red_group), dtype=torch.float32, device=X.device) if red_group else None
            _lib.check(_lib.lib().usip_mlp_gemm_x2h_red_f32(_ptr(planes), None if pool is not None else _ptr(X), _opt(X2),
                                                            _opt(coef), int(pro), _opt(pool_dp), _opt(pool_arg),
                                                            int(pool_group), y_ptr, _ptr(red[0]), _ptr(red[1]), _ptr(flat),
                                                            _opt(gsum), int(red_group), M, K, P, nb, _stream(X)),
                       "usip_mlp_gemm_x2h_red_f32")
            r = RedSums(flat, M, blocks=red_tiles)
            r.gsum = gsum
            return Y, stats, r
        if x3p:
            fn = "usip_mlp_gemm_x2h_f32" if x2h else "usip_mlp_gemm_x3p_f32"
            _lib.check(getattr(_lib.lib(), fn)(_ptr(planes), None if pool is not None else _ptr(X), _opt(X2),
                                               _opt(coef), int(pro), _opt(bias), _opt(rowbias), int(rb_group),
                                               _opt(pool_dp), _opt(pool_arg), int(pool_group), y_ptr,
                                               int(y_rows), _opt(stats), M, K, P, nb, _stream(X)), fn)
            return (Y, stats, None) if red is not None else (Y, stats)
        _lib.check(getattr(_lib.lib(), fn_name)(a_ptr, -lda if a_trans else lda,
                                                None if pool is not None else _ptr(X), _opt(X2),
                                                _opt(coef), int(pro), _opt(bias), _opt(rowbias), int(rb_group),
                                                _opt(pool_dp), _opt(pool_arg), int(pool_group),
                                                y_ptr, int(y_rows), _opt(stats), M, K, P, nb, _stream(X)), fn_name)
    return (Y, stats, None) if red is not None else (Y, stats)


def bn_finalize(stats, count, gamma, beta, eps, momentum, running_mean, running_var):
    """-> (mean [C], invstd [C], coef [2,C]); updates running_mean/var in place when given."""
    _, C, tiles = stats.shape
    dev = stats.device
    mean = torch.empty(C, dtype=torch.float32, device=dev)
    invstd = torch.empty(C, dtype=torch.float32, device=dev)
    coef = torch.empty((4, C), dtype=torch.float32, device=dev)      # (scale, shift, mean, invstd)
    with torch.cuda.device(dev), prof.kernel("bn_finalize", 4.0 * 2 * tiles * C):
        _lib.check(_lib.lib().usip_bn_finalize_f32(_ptr(stats), tiles, C, int(count), _opt(gamma), _opt(beta),
                                                   float(eps), float(momentum), _opt(running_mean),
                                                   _opt(running_var), _ptr(mean), _ptr(invstd), _ptr(coef),
                                                   _stream(stats)), "usip_bn_finalize_f32")
    # the f32x2 kernels bound |relu(bn(y))| by |gamma| sqrt(n) + |beta| with n = the samples of THEIR launch; that holds
    # when the statistics were taken over no more samples than that (bound_covers)
    declare_samples(coef, int(count))
    return mean, invstd, coef


# data_ptr of a [4,C] forward-coefficient tensor -> (the number of samples its statistics were taken over, its element
# count).  Keyed by ADDRESS, not by a Python attribute: the tensor a backward pass gets back from ctx.saved_tensors, a detach()
# or a row view is another Python object over the same memory, and an attribute would be gone.  bn_finalize
# re-declares on every call, so an address that the allocator hands out again carries its newest producer's count; a
# hand-built tensor of ANOTHER size that lands on a recycled address does not inherit the old count (the element count is
# checked: ADVICE r5), and the table sheds its OLDEST entries (a step declares a few dozen: nothing of a running step's
# forward is dropped before its backward) instead of being cleared.
import collections
_SAMPLES = collections.OrderedDict()
_SAMPLES_MAX = 8192
UNKNOWN_SAMPLE_LOOKUPS = 0          # bound_covers calls that found no recorded count (a whole step must leave this at 0)


def declare_samples(coef, count: int):
    """Record that the batch statistics in `coef` were taken over `count` samples per channel."""
    key = coef.data_ptr()
    _SAMPLES.pop(key, None)
    _SAMPLES[key] = (int(count), int(coef.numel()))           # (re-)inserted at the young end
    while len(_SAMPLES) > _SAMPLES_MAX:
        _SAMPLES.popitem(last=False)


def bound_covers(coef, samples: int) -> bool:
    """May a launch over `samples` positions per channel use the BatchNorm bound of `coef` (its [4,C] batch
    statistics)?  |y - mean| <= sqrt(n var) holds over the n samples the statistics were taken over; a launch that
    assumes n' = its own sample count under-estimates the bound when n' < n (a slice of the tensor, coefficients of
    another batch) and could overflow the fp16 planes.  Coefficients WITHOUT a recorded count do not cover anything
    (the launch takes the bound-free f32x3 path); hand-built coefficients in tests and tools either call
    declare_samples or set USIP_ASSUME_LAUNCH_SAMPLES=1 (the `assume_launch_samples` fixture of tests/conftest.py), which takes
    them as the launch's own."""
    global UNKNOWN_SAMPLE_LOOKUPS
    rec = _SAMPLES.get(coef.data_ptr())
    if rec is not None and rec[1] != int(coef.numel()) and coef._base is None:
        rec = None                  # another tensor on a recycled address (a view of the declared one keeps its base's count)
    if rec is None:
        UNKNOWN_SAMPLE_LOOKUPS += 1
        return os.environ.get("USIP_ASSUME_LAUNCH_SAMPLES") == "1"
    return rec[0] <= samples


def bn_apply(Y, coef, relu: bool):
    """Z = relu?(Y * coef[0] + coef[1]); Y [nb,C,P]."""
    nb, C, P = Y.shape
    Z = torch.empty_like(Y)
    with torch.cuda.device(Y.device), prof.kernel("bn_apply", 8.0 * nb * C * P):
        _lib.check(_lib.lib().usip_bn_apply_f32(_ptr(Y), _ptr(coef), _ptr(Z), int(bool(relu)), nb, C, P,
                                                _stream(Y)), "usip_bn_apply_f32")
    return Z


def bn_backward_reduce(dZ, Y, coef_fwd, mean, invstd, gamma, relu: bool, group: int = 0,
                       dgamma_out=None, dbeta_out=None):
    """-> (dgamma [C], dbeta [C], coef4 [4,C], gsum).  Y None: plain mode, returns (None, sum dZ, None, None).
    group > 0 additionally returns gsum [2,nb,C,P/group] (per-neighbourhood sums of dYhat and y)."""
    nb, C, P = dZ.shape
    dev = dZ.device
    # f32x2 mode: the pass also takes max |dYhat| per row and leaves a bound of |dY| in row 4 of coef4 ([5][C])
    want_bound = 1 if (_matmul_mode == "f32x2" and Y is not None) else 0
    partial = torch.empty((3 if want_bound else 2) * nb * C, dtype=torch.float32, device=dev)
    dbeta = dbeta_out if dbeta_out is not None else torch.empty(C, dtype=torch.float32, device=dev)
    dgamma = coef4 = gsum = None
    if Y is not None:
        dgamma = dgamma_out if dgamma_out is not None else torch.empty(C, dtype=torch.float32, device=dev)
        coef4 = torch.empty((5 if want_bound else 4, C), dtype=torch.float32, device=dev)
        if group:
            gsum = torch.empty((2, nb, C, P // group), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev), prof.kernel("bn_backward_reduce", 4.0 * nb * C * P * (1 if Y is None else 2)):
        _lib.check(_lib.lib().usip_bn_backward_reduce_f32(_ptr(dZ), _opt(Y), _opt(coef_fwd), _opt(mean), _opt(invstd),
                                                          _opt(gamma), int(bool(relu)), _ptr(partial), _opt(dgamma),
                                                          _ptr(dbeta), _opt(coef4), _opt(gsum), int(group),
                                                          nb, C, P, want_bound, _stream(dZ)),
                   "usip_bn_backward_reduce_f32")
    return dgamma, dbeta, coef4, gsum


# Deferred weight-gradient reductions (include/usip_hip.h: usip_wgrad_defer / usip_wgrad_flush): inside a training step the
# fixed-order sums of partial tiles are recorded by the library and issued together after backward.  The partial-tile
# workspaces must outlive the flush, so while the mode is on every workspace allocated here is kept in _DEFER_KEEP.
_DEFER_KEEP = None


def _keep(ws):
    if _DEFER_KEEP is not None:
        _DEFER_KEEP.append(ws)
    return ws


def wgrad_defer(on: bool, device=None):
    """Enter (True) or leave (False) the deferred mode; leaving drops whatever is recorded and not flushed.
    device: record only what is launched on that device's CURRENT stream (the step's); anything another thread, device or
    stream launches meanwhile runs its reduction at once (the library's job list is process-global)."""
    global _DEFER_KEEP
    if on and device is not None:
        dev = torch.device(device)
        _lib.check(_lib.lib().usip_wgrad_defer_on(ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
                   "usip_wgrad_defer_on")
    else:
        _lib.check(_lib.lib().usip_wgrad_defer(1 if on else 0), "usip_wgrad_defer")
    _DEFER_KEEP = [] if on else None


class _reduce_now:
    """`with _reduce_now(dest_is_private):` -- a weight gradient whose destination is NOT handed in by the caller (no view of
    the step's gradient bucket: a frozen sibling parameter switched the layer's sink off, functional._sink) goes back to
    autograd, whose AccumulateGrad reads it before any flush: its fixed-order sum must be launched at once."""

    def __init__(self, private: bool):
        self.on = bool(private) and _DEFER_KEEP is not None

    def __enter__(self):
        if self.on:
            self.was = int(_lib.lib().usip_wgrad_defer_hold(1))

    def __exit__(self, *exc):
        if self.on:
            _lib.lib().usip_wgrad_defer_hold(self.was)
        return False


def wgrad_flush(device) -> int:
    """Issue every recorded reduction on `device`'s current stream (two launches at most); -> how many were issued.
    The mode stays on; the workspaces are released (stream-ordered: their memory is reused only behind the flush)."""
    global _DEFER_KEEP
    dev = torch.device(device)
    with torch.cuda.device(dev), prof.kernel("wgrad_reduce_all", 0.0):
        n = int(_lib.lib().usip_wgrad_flush(ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    if n < 0:
        raise RuntimeError("usip_amd: usip_wgrad_flush failed (%d)" % n)
    if _DEFER_KEEP is not None:
        _DEFER_KEEP = []
    return n


def bn_pool_backward_reduce(dpooled, arg, Y4, coef_fwd, mean, invstd, gamma, relu: bool,
                            dgamma_out=None, dbeta_out=None, yarg=None):
    """BN(+ReLU) backward sums for a layer that fed only a max over K: -> (dgamma, dbeta, coef4).
    yarg: Y4 at the arg-max as group_max_act(..., want_yarg=True) returned it (saves the gathers)."""
    nb, C, M, K = Y4.shape
    dev = Y4.device
    want_bound = 1 if _matmul_mode == "f32x2" else 0
    partial = torch.empty((3 if want_bound else 2) * nb * C, dtype=torch.float32, device=dev)
    dbeta = dbeta_out if dbeta_out is not None else torch.empty(C, dtype=torch.float32, device=dev)
    dgamma = dgamma_out if dgamma_out is not None else torch.empty(C, dtype=torch.float32, device=dev)
    coef4 = torch.empty((5 if want_bound else 4, C), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev), prof.kernel("bn_backward_reduce_pooled", 4.0 * nb * C * M * 3):
        _lib.check(_lib.lib().usip_bn_pool_backward_reduce_f32(_ptr(dpooled), _ptr(arg), _ptr(Y4), _opt(yarg),
                                                               _ptr(coef_fwd), _ptr(mean), _ptr(invstd), _opt(gamma),
                                                               int(bool(relu)), _ptr(partial), _ptr(dgamma), _ptr(dbeta),
                                                               _ptr(coef4), nb, C, M, K, want_bound, _stream(Y4)),
                   "usip_bn_pool_backward_reduce_f32")
    return dgamma, dbeta, coef4


def mlp_wgrad(G, X, pro: int = 0, G2=None, coef4=None, out=None, coloff: int = 0, xcoef=None, pool=None):
    """dW [M,N] = sum_{b,p} pro(G)[b,m,p] * X[b,n,p]; G [nb,M,P], X [nb,N,P].
    With `out` ([M, ldw] contiguous) the result is written into columns [coloff, coloff+N) of it."""
    pool_dp = pool_arg = None
    pool_group = 0
    if pool is not None:                      # pro == 3: G is not a tensor
        pool_dp, pool_arg, pool_group = pool
        nb, M, P = G2.shape
    else:
        nb, M, P = G.shape
    N = X.shape[1]
    dev = X.device
    ws_n = _lib.lib().usip_mlp_wgrad_workspace(M, N, P, nb)
    ws = _keep(torch.empty(max(int(ws_n), 1), dtype=torch.float32, device=dev))
    dW = out if out is not None else torch.empty((M, N), dtype=torch.float32, device=dev)
    ldw = dW.shape[1]
    bf16 = _matmul_mode == "bf16"
    fn_name = {"bf16": "usip_mlp_wgrad_bf16", "f32x3": "usip_mlp_wgrad_f32x3",
               "f32x2": "usip_mlp_wgrad_f32x3"}.get(_matmul_mode, "usip_mlp_wgrad_f32")
    x3 = _x3_family() and not (M <= 64 and N <= 64) and bool(_lib.lib().usip_mlp_wgrad_f32x3_used(M, N, P, nb))
    # two fp16 planes: both operands need a bound (coef4 with its fifth row, xcoef = batch statistics)
    x2h = (x3 and _matmul_mode == "f32x2" and pro >= 2 and coef4 is not None and coef4.shape[0] >= 5 and M > 128 and N > 128
           and xcoef is not None and xcoef.shape[0] >= 4 and P % 4 == 0 and bound_covers(xcoef, nb * P))
    if x2h:
        fn_name = "usip_mlp_wgrad_x2h_f32"

    def _key():
        t = 1 if (M <= 64 and N <= 64) else 2
        if x3:
            blocks = _lib.lib().usip_mlp_wgrad_f32x3_blocks(M, N, P, nb)
            if blocks < 0:
                if x2h and not (_lib.lib().usip_tuning_value(7) & 1) and max(M, N) * P < (1 << 30):
                    return "wgrad_x2l_kernel<%d> |wg=%d" % (pro, -blocks)         # round 5: the full-line form
                return "wgrad_x3_kernel<%d, %s, %d> |wg=%d" % (pro, "true" if xcoef is not None else "false",
                                                               2 if x2h else 3, -blocks)
        planes = ", 1" if bf16 else (", 3" if x3 else "")
        return "%s<%d, %d, %d, %s, %s%s> |wg=%d" % ("wgrad_bf16_kernel" if (bf16 or x3) else "wgrad_kernel", t, t, pro,
                                                    "true" if xcoef is not None else "false",
                                                    "true" if P % 4 == 0 else "false", planes,
                                                    _lib.lib().usip_mlp_wgrad_blocks(M, N, P, nb))

    with torch.cuda.device(dev), _reduce_now(out is None), prof.kernel(
            "shared_mlp_wgrad %dx%d" % (M, N), 4.0 * nb * P * (M * (2 if pro == 2 else 1) + N), 2.0 * M * N * nb * P,
            rocprof_key=_key):
        _lib.check(getattr(_lib.lib(), fn_name)(_opt(G), _opt(G2), _opt(coef4), int(pro), _ptr(X), _opt(xcoef),
                                                _opt(pool_dp), _opt(pool_arg), int(pool_group), _ptr(ws), _ptr(dW),
                                                int(ldw), int(coloff), M, N, P, nb, _stream(X)), fn_name)
    return dW


reduce_now = _reduce_now          # for callers that hand a PRIVATE destination through `out` / `dw_out` (functional.py)


class RedSums:
    """What a fused layer backward leaves for the layer that produced its input: `sums` [2, blocks, C] partial
    BatchNorm-backward sums and `maxima` [blocks] of |dX [relu on]| -- views of one flat buffer."""

    def __init__(self, flat: torch.Tensor, C: int, blocks: Optional[int] = None):
        """blocks: rows of the partial sums when the buffer holds another number of maxima than that (the direct GEMM's
        data-gradient epilogue: one maximum per 256-row tile)."""
        if blocks is None:
            blocks = flat.numel() // (2 * C + 1)
        self.flat = flat
        self.sums = flat[:2 * blocks * C].view(2, blocks, C)
        self.maxima = flat[2 * blocks * C:]
        self.gsum = None                     # per-neighbourhood sums [2,nb,C,G], when the producer took them too
        self.wsum = None                     # (wsum, wsum3, source tensor): the producing layer's weight-gradient sums (WSUM)


def narrow_backward_supported(Cin: int, Cout: int, P: int, tensors=()) -> bool:
    """True when usip_mlp_narrow_backward_f32 takes this layer.  `tensors`: the (dZ, Y, X) the call would pass --
    the kernel needs them 16-byte aligned (an offset view handed over by autograd is not) and addresses elements
    with 32-bit offsets; anything else goes through the generic data- and weight-gradient kernels instead of
    raising in the middle of a backward."""
    if _matmul_mode == "bf16" or not _lib.lib().usip_mlp_narrow_backward_supported(int(Cin), int(Cout), int(P)):
        return False
    for t in tensors:
        if t is None:
            continue
        if t.data_ptr() % 16 != 0 or t.numel() >= 2 ** 32 or not t.is_contiguous():
            return False
    return True


def mlp_narrow_backward(dz, y, coef4, x, xcoef, w2, wcol: int = 0, dw_out=None, Cin: int = 64, want_red: bool = False):
    """Fused backward of a narrow layer (csrc/narrow_bwd.hip): -> (dx [nb,Cin,P], dW[, red]).
    dz, y [nb,Cout,P]; x [nb,Cin,P]; w2 [Cout, Ctot] contiguous, the layer's inputs are its columns [wcol, wcol+Cin);
    dw_out ([Cout, Ctot] contiguous) receives the weight gradient in the same columns.
    want_red (xcoef = the producing layer's [4,Cin] forward coefficients): also returns a RedSums: partial
    BatchNorm-backward sums [2, blocks, Cin] of the producing layer against dx and the workgroups' maxima of
    |dx [relu on]| (see bn_backward_from_partials)."""
    nb, Cout, P = dz.shape
    dev = dz.device
    for t, n in ((dz, "dz"), (y, "y"), (x, "x"), (w2, "w2"), (coef4, "coef4")):
        _need(t, n, torch.float32)
    ldw = w2.shape[1]
    dW = dw_out if dw_out is not None else torch.empty_like(w2)
    dx = torch.empty((nb, Cin, P), dtype=torch.float32, device=dev)
    ws = _keep(torch.empty(int(_lib.lib().usip_mlp_narrow_backward_workspace(Cout, P, nb)), dtype=torch.float32, device=dev))
    red = None
    if want_red:
        if xcoef is None or xcoef.shape[0] < 4:
            raise RuntimeError("mlp_narrow_backward: want_red needs the producing layer's [4, Cin] coefficients")
        blocks = int(_lib.lib().usip_mlp_narrow_backward_blocks(Cout, P, nb))
        red = torch.empty(2 * blocks * Cin + blocks, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev), _reduce_now(dw_out is None), prof.kernel("shared_mlp_narrow_bwd %dx%d" % (Cout, Cin),
                                             4.0 * nb * P * (2 * Cout + 2 * Cin), 4.0 * Cout * Cin * nb * P,
                                             rocprof_key="narrow_bwd_kernel<%d, %s, %s> |wg=%d" % (
                                                 Cout, "true" if xcoef is not None else "false",
                                                 "true" if want_red else "false", ws.numel() // (Cout * 64))):
        _lib.check(_lib.lib().usip_mlp_narrow_backward_f32(
            _ptr(dz), _ptr(y), _ptr(coef4), _ptr(x), int(x.shape[1]), _opt(xcoef),
            ctypes.c_void_p(w2.data_ptr() + 4 * int(wcol)), int(ldw), _ptr(dx), Cin, _ptr(ws),
            ctypes.c_void_p(dW.data_ptr() + 4 * int(wcol)), int(dW.shape[1]), _opt(red), Cin, Cout, P, nb, _stream(dz)),
            "usip_mlp_narrow_backward_f32")
    return (dx, dW, RedSums(red, Cin)) if want_red else (dx, dW)


def layer_backward_x2_supported(Cin: int, Cout: int, P: int, tensors=(), coef4=None, xcoef=None, pooled: bool = False) -> bool:
    """True when usip_mlp_layer_backward_x2h_f32 takes this layer: f32x2 mode, a supported shape, the bounds the two
    fp16 planes need (coef4 with its fifth row, the producing layer's four-row coefficients), aligned contiguous tensors."""
    if _matmul_mode != "f32x2" or not LAYER_BWD_X2:
        return False
    if coef4 is None or coef4.shape[0] < 5 or xcoef is None or xcoef.shape[0] < 4:
        return False
    nb = next((t.shape[0] for t in tensors if t is not None and t.dim() == 3), None)
    if nb is not None and not bound_covers(xcoef, int(nb) * int(P)):
        return False
    if not _lib.lib().usip_mlp_layer_backward_x2h_supported(int(Cin), int(Cout), int(P), 1 if pooled else 0):
        return False
    for t in tensors:
        if t is None:
            continue
        if t.data_ptr() % 16 != 0 or t.numel() >= 2 ** 32 or not t.is_contiguous():
            return False
    return True


# f32x2 mode: the fused layer backward on the 16-bit matrix cores; USIP_LAYER_BWD_X2=0 for A/B runs
LAYER_BWD_X2 = _os.environ.get("USIP_LAYER_BWD_X2", "1") not in ("0", "off")


def mlp_layer_backward_x2(dz, y, coef4, x, xcoef, w2, wcol: int = 0, dw_out=None, Cin: int = 64, want_red: bool = False,
                          pool=None, want_gsum: bool = False, wsrc=None):
    """Fused backward of a <= 128-wide layer with f32x2 products (csrc/layer_bwd_x2.hip): -> (dx [nb,Cin,P], dW[, red]).
    Arguments as mlp_narrow_backward; coef4 [5,Cout] (with bounds), xcoef [4,Cin]; pool = (dpooled [nb,Cout,G],
    arg i32 [nb,Cout,G], group) with dz None for the pooled form.  red: a RedSums (partial sums [2, blocks, Cin] and the
    workgroups' maxima of |dx [relu on]|); want_gsum (pooled form with red, group % 32 == 0): red.gsum [2,nb,Cin,G] =
    the per-neighbourhood sums of dx [relu on] and of x that bn_backward_reduce(group=...) would return.
    wsrc ([nb, <= 8, P], the gradient-free INPUT of the layer that produced x; (64, 64) form with want_red): red.wsum = the
    sums that layer's weight gradient follows from (wsum_finalize) -- it then needs no pass over its own (dZ, Y)."""
    nb, Cout, P = y.shape
    dev = y.device
    for t, n in ((y, "y"), (x, "x"), (w2, "w2"), (coef4, "coef4"), (xcoef, "xcoef")):
        _need(t, n, torch.float32)
    ldw = w2.shape[1]
    dW = dw_out if dw_out is not None else torch.empty_like(w2)
    dx = torch.empty((nb, Cin, P), dtype=torch.float32, device=dev)
    lib = _lib.lib()
    blocks = int(lib.usip_mlp_layer_backward_x2h_blocks(Cin, Cout, P, nb))
    ws = _keep(torch.empty(int(lib.usip_mlp_layer_backward_x2h_workspace(Cin, Cout, P, nb)), dtype=torch.float32, device=dev))
    red = torch.empty(2 * blocks * Cin + blocks, dtype=torch.float32, device=dev) if want_red else None
    planes = weight_planes(w2, wcol, Cin, Cout, 2, P, nb)      # W as the data-gradient operand: K-major [Cout][ldw]
    pdp = parg = gsum = None
    group = 0
    if pool is not None:
        pdp, parg, group = pool
        _need(pdp, "dpooled", torch.float32)
        _need(parg, "arg", torch.int32)
        if want_gsum:
            if not want_red or group % 32:
                raise ValueError("mlp_layer_backward_x2: group sums need want_red and a group that is a multiple of 32")
            gsum = torch.empty((2, nb, Cin, P // group), dtype=torch.float32, device=dev)
    else:
        _need(dz, "dz", torch.float32)
    pg = pool is not None and group % 32 == 0
    wsum = wsum3 = None
    if want_red and wsum_supported(wsrc, Cin, Cout, P, pool is not None):
        wsum = torch.empty((blocks, Cin, 16), dtype=torch.float32, device=dev)
        wsum3 = torch.empty((blocks, 8), dtype=torch.float32, device=dev)
    key = ("layer_bwd_x2ws_kernel<%d, %d, 4, 32> |wg=%d" % (Cin, Cout, blocks)) if wsum is not None else (
        "layer_bwd_x2_kernel<%d, %d, %s, %s, %d, 32, %s, %s, %s> |wg=%d" % (
            Cin, Cout, "true" if pool is not None else "false", "true" if want_red else "false", 8 if Cin == 128 else 4,
            "true" if (Cin == 128 and not (want_red and pg)) else "false", "true" if pg else "false",
            "true" if (pg or (Cin == 64 and Cout == 128)) else "false", blocks))
    with torch.cuda.device(dev), _reduce_now(dw_out is None), prof.kernel(
            "shared_mlp_layer_bwd_x2%s %dx%d" % ("ws" if wsum is not None else "", Cout, Cin),
            4.0 * nb * P * ((1 if pool is not None else 2) * Cout + 2 * Cin + (wsrc.shape[1] if wsum is not None else 0)),
            4.0 * Cout * Cin * nb * P, rocprof_key=key):
        if wsum is not None:
            _lib.check(lib.usip_mlp_layer_backward_x2h_ws_f32(
                _ptr(dz), _ptr(y), _ptr(coef4), _ptr(x), int(x.shape[1]), _ptr(xcoef), ctypes.c_void_p(planes.data_ptr()),
                _ptr(dx), Cin, _ptr(ws), ctypes.c_void_p(dW.data_ptr() + 4 * int(wcol)), int(dW.shape[1]), _ptr(red),
                _ptr(wsrc), int(wsrc.shape[1]), _ptr(wsum), _ptr(wsum3), Cin, Cout, P, nb, _stream(y)),
                "usip_mlp_layer_backward_x2h_ws_f32")
        else:
            _lib.check(lib.usip_mlp_layer_backward_x2h_f32(
                _opt(dz), _ptr(y), _ptr(coef4), _opt(pdp), _opt(parg), int(group), _ptr(x), int(x.shape[1]), _ptr(xcoef),
                ctypes.c_void_p(planes.data_ptr()), _ptr(dx), Cin, _ptr(ws), ctypes.c_void_p(dW.data_ptr() + 4 * int(wcol)),
                int(dW.shape[1]), _opt(red), _opt(gsum), Cin, Cout, P, nb, _stream(y)), "usip_mlp_layer_backward_x2h_f32")
    if not want_red:
        return dx, dW
    r = RedSums(red, Cin)
    r.gsum = gsum
    if wsum is not None:
        r.wsum = (wsum, wsum3, wsrc)
    return dx, dW, r


WSUM = _os.environ.get("USIP_WSUM", "1") not in ("0", "off")        # A/B switch of the fused first-layer weight gradient


def wsum_supported(wsrc, Cin: int, Cout: int, P: int, pooled: bool) -> bool:
    """May mlp_layer_backward_x2 take the producing layer's weight-gradient sums against `wsrc` on the way?"""
    return (WSUM and wsrc is not None and not pooled and Cin == 64 and Cout == 64 and wsrc.dim() == 3 and wsrc.shape[1] <= 8
            and wsrc.shape[2] == P and wsrc.is_contiguous() and wsrc.data_ptr() % 16 == 0 and wsrc.dtype == torch.float32)


def wsum_finalize(wsum_pack, coef4, mean, dW):
    """dW[:, :rows] = the weight gradient of the layer whose sums `wsum_pack` (RedSums.wsum) holds, from its own coef4 /
    batch mean (csrc/layer_bwd_x2.hip::wsum_finalize_kernel); dW [C, >= rows] contiguous rows."""
    wsum, wsum3, wsrc = wsum_pack
    blocks = wsum3.shape[0]
    C = wsum.shape[1]
    with torch.cuda.device(dW.device), prof.kernel("wsum_finalize", 4.0 * blocks * (16 * C + 8)):
        _lib.check(_lib.lib().usip_mlp_wsum_finalize_f32(_ptr(wsum), _ptr(wsum3), int(blocks), int(C), _ptr(coef4), _ptr(mean),
                                                         int(wsrc.shape[1]), _ptr(dW), int(dW.stride(0)), _stream(dW)),
                   "usip_mlp_wsum_finalize_f32")
    return dW


def bn_pool_backward_partials(dpooled, arg, Y4, coef_fwd, mean, invstd, relu: bool, yarg=None):
    """Partial BatchNorm-backward sums [2, nb, C] of a gradient that is dpooled at the arg-max positions and zero
    elsewhere (the sparse half of a layer output that feeds a max-pool AND another layer).  In f32x2 mode a RedSums
    whose maxima [nb * C] are those of |dpooled [relu on]| per cloud and channel."""
    nb, C, M, K = Y4.shape
    want_max = _matmul_mode == "f32x2"
    partial = torch.empty((3 if want_max else 2) * nb * C, dtype=torch.float32, device=Y4.device)
    with torch.cuda.device(Y4.device), prof.kernel("bn_backward_reduce_pooled", 4.0 * nb * C * M * 3):
        _lib.check(_lib.lib().usip_bn_pool_backward_reduce_f32(_ptr(dpooled), _ptr(arg), _ptr(Y4), _opt(yarg),
                                                               _ptr(coef_fwd), _ptr(mean), _ptr(invstd), None,
                                                               int(bool(relu)), _ptr(partial), None, None, None, nb, C,
                                                               M, K, 1 if want_max else 0, _stream(Y4)),
                   "usip_bn_pool_backward_reduce_f32")
    if want_max:
        r = RedSums.__new__(RedSums)
        r.flat, r.sums, r.maxima = partial, partial[:2 * nb * C].view(2, nb, C), partial[2 * nb * C:]
        return r
    return partial.view(2, nb, C)


def bn_backward_from_partials(partials, count: int, coef_fwd, mean, invstd, dgamma_out=None, dbeta_out=None):
    """(dgamma, dbeta, coef4) from partial sums [2, rows, C] (one tensor / RedSums or a list of them: all are summed).
    When every part is a RedSums (at most two) coef4 gets its fifth row, the bound of |dY| the f32x2 kernels need:
    |dYhat| <= the sum of the parts' maxima."""
    parts = list(partials) if isinstance(partials, (list, tuple)) else [partials]
    maxima = [p.maxima for p in parts] if all(isinstance(p, RedSums) for p in parts) and len(parts) <= 2 else None
    sums = [p.sums if isinstance(p, RedSums) else p for p in parts]
    sums = torch.cat(sums, dim=1).contiguous() if len(sums) > 1 else sums[0]
    _, rows, C = sums.shape
    dev = sums.device
    dgamma = dgamma_out if dgamma_out is not None else torch.empty(C, dtype=torch.float32, device=dev)
    dbeta = dbeta_out if dbeta_out is not None else torch.empty(C, dtype=torch.float32, device=dev)
    coef4 = torch.empty((5 if maxima else 4, C), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev), prof.kernel("bn_backward_finalize", 8.0 * rows * C):
        if maxima:
            m1 = maxima[1] if len(maxima) > 1 else None
            _lib.check(_lib.lib().usip_bn_backward_finalize_max_f32(
                _ptr(sums), int(rows), int(C), int(count), _ptr(coef_fwd), _ptr(mean), _ptr(invstd), _ptr(dgamma),
                _ptr(dbeta), _ptr(coef4), _ptr(maxima[0]), int(maxima[0].numel()), _opt(m1),
                0 if m1 is None else int(m1.numel()), _stream(sums)), "usip_bn_backward_finalize_max_f32")
        else:
            _lib.check(_lib.lib().usip_bn_backward_finalize_f32(_ptr(sums), int(rows), int(C), int(count), _ptr(coef_fwd),
                                                                _ptr(mean), _ptr(invstd), _ptr(dgamma), _ptr(dbeta),
                                                                _ptr(coef4), _stream(sums)),
                       "usip_bn_backward_finalize_f32")
    return dgamma, dbeta, coef4


# --------------------------------------------------------------------------- grouping / pooling
def group_gather(x, idx32, sub=None, out=None, coff: int = 0):
    """out[b, coff+c, m, k] = x[b, c, idx[b,m,k]] - (c < nsub ? sub[b,c,m] : 0).  x [B,C,N], idx i32 [B,M,K],
    sub [B,nsub,M] or None.  `out` [B,Ctot,M,K] lets several gathers fill one pre-concatenated tensor."""
    _need(x, "x", torch.float32)
    _need(idx32, "idx", torch.int32)
    B, C, N = x.shape
    _, M, K = idx32.shape
    if out is None:
        out = torch.empty((B, C, M, K), dtype=torch.float32, device=x.device)
    Ctot = out.shape[1]
    nsub = 0 if sub is None else sub.shape[1]
    with torch.cuda.device(x.device), prof.kernel("group_gather", 4.0 * B * M * K * (C + 1)):
        _lib.check(_lib.lib().usip_group_gather_f32(_ptr(x), _ptr(idx32), _opt(sub), _ptr(out), B, C, N, M, K, nsub,
                                                    Ctot, int(coff), _stream(x)), "usip_group_gather_f32")
    return out


def group_gather_backward(dout, idx32, C: int, N: int, coff: int = 0):
    """dx [B,C,N] = scatter-add of dout[:, coff:coff+C] (dout [B,Ctot,M,K] contiguous)."""
    _need(dout, "dout", torch.float32)
    B, Ctot, M, K = dout.shape
    dx = torch.empty((B, C, N), dtype=torch.float32, device=dout.device)
    with torch.cuda.device(dout.device), prof.kernel("group_gather_bwd", 4.0 * B * M * K * (C + 1)):
        _lib.check(_lib.lib().usip_group_gather_backward_f32(_ptr(dout), _ptr(idx32), _ptr(dx), B, C, N, M, K, Ctot,
                                                             int(coff), _stream(dout)), "usip_group_gather_backward_f32")
    return dx


def group_max(z):
    """z [B,C,M,K] -> (pooled [B,C,M], arg i32 [B,C,M])."""
    _need(z, "z", torch.float32)
    B, C, M, K = z.shape
    pooled = torch.empty((B, C, M), dtype=torch.float32, device=z.device)
    arg = torch.empty((B, C, M), dtype=torch.int32, device=z.device)
    with torch.cuda.device(z.device), prof.kernel("group_max", 4.0 * B * C * M * (K + 2)):
        _lib.check(_lib.lib().usip_group_max_f32(_ptr(z), _ptr(pooled), _ptr(arg), B * C * M, K, _stream(z)),
                   "usip_group_max_f32")
    return pooled, arg


def group_max_act(y4, coef, relu: bool, want_yarg: bool = False):
    """max_k relu?(y*coef[0]+coef[1]) straight from a layer's pre-BN output y4 [B,C,M,K] -> (pooled, arg[, yarg]);
    yarg = y4 at the arg-max."""
    _need(y4, "y", torch.float32)
    B, C, M, K = y4.shape
    pooled = torch.empty((B, C, M), dtype=torch.float32, device=y4.device)
    arg = torch.empty((B, C, M), dtype=torch.int32, device=y4.device)
    yarg = torch.empty((B, C, M), dtype=torch.float32, device=y4.device) if want_yarg else None
    with torch.cuda.device(y4.device), prof.kernel("group_max", 4.0 * B * C * M * (K + 2)):
        _lib.check(_lib.lib().usip_group_max_act_f32(_ptr(y4), _ptr(coef), int(bool(relu)), _ptr(pooled), _ptr(arg),
                                                     _opt(yarg), B, C, M, K, _stream(y4)), "usip_group_max_act_f32")
    return (pooled, arg, yarg) if want_yarg else (pooled, arg)


def group_max_backward(dpooled, arg, K: int):
    _need(dpooled, "dpooled", torch.float32)
    B, C, M = dpooled.shape
    dz = torch.empty((B, C, M, int(K)), dtype=torch.float32, device=dpooled.device)
    with torch.cuda.device(dpooled.device), prof.kernel("group_max_bwd", 4.0 * B * C * M * (K + 2)):
        _lib.check(_lib.lib().usip_group_max_backward_f32(_ptr(dpooled), _ptr(arg), _ptr(dz), B * C * M, int(K),
                                                          _stream(dpooled)), "usip_group_max_backward_f32")
    return dz


def group_max_backward_add_(dz, dpooled, arg):
    """In place: dz[b,c,m,arg[b,c,m]] += dpooled[b,c,m]; returns dz."""
    _need(dz, "dz", torch.float32)
    _need(dpooled, "dpooled", torch.float32)
    _need(arg, "arg", torch.int32)
    B, C, M, K = dz.shape
    if dpooled.shape != (B, C, M) or arg.shape != (B, C, M):
        raise RuntimeError("group_max_backward_add_: shapes do not match")
    with torch.cuda.device(dz.device), prof.kernel("group_max_bwd_add", 4.0 * B * C * M * 4):
        _lib.check(_lib.lib().usip_group_max_backward_add_f32(_ptr(dpooled), _ptr(arg), _ptr(dz), B * C * M, int(K),
                                                              _stream(dz)), "usip_group_max_backward_add_f32")
    return dz


def multi_transpose(src_flat, dst_flat, table, total_tiles: int):
    """K-major copies of all weight matrices described by `table` (int32 [n,5]: source offset, rows, cols,
    destination offset, first tile) from src_flat into dst_flat, one launch."""
    _need(src_flat, "src", torch.float32)
    _need(dst_flat, "dst", torch.float32)
    _need(table, "table", torch.int32)
    with torch.cuda.device(src_flat.device), prof.kernel("weight_transposes", 8.0 * dst_flat.numel()):
        _lib.check(_lib.lib().usip_multi_transpose_f32(_ptr(src_flat), _ptr(dst_flat), _ptr(table), int(table.shape[0]),
                                                       int(total_tiles), _stream(src_flat)), "usip_multi_transpose_f32")
    return dst_flat


def nearest_nd(a: torch.Tensor, b: torch.Tensor):
    """f-1: (min_j |a_i - b_j| [B,Ma], first arg-min i32 [B,Ma]) for C-dimensional a [B,C,Ma], b [B,C,Nb<=1024]."""
    _need(a, "a", torch.float32)
    _need(b, "b", torch.float32)
    B, C, Ma = a.shape
    Nb = b.shape[2]
    d = torch.empty((B, Ma), dtype=torch.float32, device=a.device)
    arg = torch.empty((B, Ma), dtype=torch.int32, device=a.device)
    with torch.cuda.device(a.device), prof.kernel("nearest_nd", 4.0 * B * C * (Ma + Nb), 3.0 * B * C * Ma * Nb):
        _lib.check(_lib.lib().usip_nearest_nd_f32(_ptr(a), _ptr(b), _ptr(d), _ptr(arg), B, C, Ma, Nb, _stream(a)),
                   "usip_nearest_nd_f32")
    return d, arg


def nearest_backward(a, b, d, arg32, gd, need_gb: bool):
    """-> (ga [B,C,Ma], gb [B,C,Nb] or None)."""
    B, C, Ma = a.shape
    Nb = b.shape[2]
    ga = torch.empty_like(a)
    gb = torch.empty_like(b) if need_gb else None
    with torch.cuda.device(a.device), prof.kernel("nearest_bwd", 4.0 * B * Ma * 12):
        _lib.check(_lib.lib().usip_nearest_backward_f32(_ptr(a), _ptr(b), _ptr(d), _ptr(arg32), _ptr(gd), _ptr(ga),
                                                        _opt(gb), B, C, Ma, Nb, _stream(a)),
                   "usip_nearest_backward_f32")
    return ga, gb


def knn(query: torch.Tensor, database: torch.Tensor, K: int) -> torch.Tensor:
    """a-7: i32 [B,M,K] nearest-first neighbour indices; query [B,3,M], database [B,3,N], N <= 1024."""
    _need_pts(query, "query")
    _need_pts(database, "database")
    B, _, M = query.shape
    N = database.shape[2]
    out = torch.empty((B, M, int(K)), dtype=torch.int32, device=query.device)
    with torch.cuda.device(query.device), prof.kernel("knn", 4.0 * B * (3 * (M + N) + M * K), 8.0 * B * M * N):
        _lib.check(_lib.lib().usip_knn_f32(_ptr(query), _ptr(database), _ptr(out), B, M, N, int(K), _stream(query)),
                   "usip_knn_f32")
    return out


def knn_layer_supported(N: int, M: int, K: int) -> bool:
    return bool(_lib.lib().usip_knn_layer_supported(int(N), int(M), int(K)))


def knn_layer_forward(U, weight2, database, query, idx32, want_stats: bool = True):
    """Y[b,co,m,k] = W[co,:3] . (database[b,:,n] - query[b,:,m]) + U[b,co,n], n = idx[b,m,k] (csrc/knn_layer.hip).
    U [B,Cout,N] (= W[:,3:] . feat + bias), weight2 [Cout,3+C] -> (Y [B,Cout,M*K], stats [2*Cout*B] or None)."""
    _need(U, "U", torch.float32)
    _need(weight2, "weight", torch.float32)
    _need_pts(query, "query")
    _need_pts(database, "database")
    _need(idx32, "idx", torch.int32)
    B, Cout, N = U.shape
    _, M, K = idx32.shape
    Y = torch.empty((B, Cout, M * K), dtype=torch.float32, device=U.device)
    stats = torch.empty(2 * Cout * B, dtype=torch.float32, device=U.device) if want_stats else None
    with torch.cuda.device(U.device), prof.kernel("knn_layer_fwd", 4.0 * B * (M * K * (Cout + 1) + Cout * N)):
        _lib.check(_lib.lib().usip_knn_layer_forward_f32(_ptr(U), _ptr(weight2), int(weight2.stride(0)), _ptr(database),
                                                         _ptr(query), _ptr(idx32), _ptr(Y), _opt(stats), B, Cout, N, M, K,
                                                         _stream(U)), "usip_knn_layer_forward_f32")
    return Y, stats


def knn_layer_backward(dZ, Y, coef4, relu: bool, dcoord, start, perm, M: int, K: int):
    """-> (dU [B,Cout,N], dWc [Cout,3]): segment sums of dY = BN'(dZ, Y) in CSR list order, and the gradient of the three
    coordinate columns of the weight (per-cloud partials added in cloud order).  dcoord [B,3,M*K] = the decentered
    neighbour coordinates (group_gather(database, idx, sub=query))."""
    _need(dZ, "dZ", torch.float32)
    _need(Y, "Y", torch.float32)
    _need(dcoord, "dcoord", torch.float32)
    B, Cout, P = Y.shape
    N = start.shape[1] - 1
    dU = torch.empty((B, Cout, N), dtype=torch.float32, device=Y.device)
    part = torch.empty((B, Cout, 3), dtype=torch.float32, device=Y.device)
    with torch.cuda.device(Y.device), prof.kernel("knn_layer_bwd", 4.0 * B * (P * (2 * Cout + 3) + Cout * N)):
        _lib.check(_lib.lib().usip_knn_layer_backward_f32(_ptr(dZ), _ptr(Y), _ptr(coef4), int(bool(relu)), _ptr(dcoord),
                                                          _ptr(start), _ptr(perm), _ptr(dU), _ptr(part), B, Cout, N,
                                                          int(M), int(K), _stream(Y)), "usip_knn_layer_backward_f32")
    return dU, part.sum(dim=0)


def adam_step(param, grad, exp_avg, exp_avg_sq, step, lr: float, beta1: float, beta2: float, eps: float, hyper=None):
    """One Adam step on flat fp32 buffers, in place (usip_adam_step_f32); `step` is a device float[1], incremented.
    hyper: device float[4] = (lr, beta1, beta2, eps) read by the kernel instead of the scalar arguments (a captured launch
    then follows a changed learning rate)."""
    for t, n in ((param, "param"), (grad, "grad"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq"), (step, "step")):
        _need(t, n, torch.float32)
    n = param.numel()
    if hyper is not None:
        _need(hyper, "hyper", torch.float32)
        with torch.cuda.device(param.device), prof.kernel("adam", 4.0 * 7 * n):
            _lib.check(_lib.lib().usip_adam_step_hyper_f32(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq),
                                                           _ptr(step), _ptr(hyper), n, _stream(param)),
                       "usip_adam_step_hyper_f32")
        return
    with torch.cuda.device(param.device), prof.kernel("adam", 4.0 * 7 * n):
        _lib.check(_lib.lib().usip_adam_step_f32(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), _ptr(step),
                                                 float(lr), float(beta1), float(beta2), float(eps), n, _stream(param)),
                   "usip_adam_step_f32")


def knn_points(node: torch.Tensor, x: torch.Tensor, K: int) -> torch.Tensor:
    """RPN_Detector_KNN's neighbourhoods (models/networks.py:576-581): i32 [B,M,K], the K cloud points nearest to every
    node, nearest first, ties towards the lower index.  node [B,3,M], x [B,3,N], N <= 16384, K <= min(N, 256)."""
    _need_pts(node, "node")
    _need_pts(x, "x")
    B, _, M = node.shape
    N = x.shape[2]
    out = torch.empty((B, M, K), dtype=torch.int32, device=x.device)
    with torch.cuda.device(x.device), prof.kernel("knn_points", 4.0 * B * (3 * N + 3 * M + M * K)):
        _lib.check(_lib.lib().usip_knn_points_f32(_ptr(node), _ptr(x), _ptr(out), B, M, N, int(K), _stream(x)),
                   "usip_knn_points_f32")
    return out


def fps(pts: torch.Tensor, first_idx: torch.Tensor, k: int) -> torch.Tensor:
    """f-3: farthest-point sampling. pts f32 [B,3,n], first_idx i32 [B] -> i32 [B,k] indices."""
    _need_pts(pts, "pts")
    _need(first_idx, "first_idx", torch.int32)
    B, _, n = pts.shape
    out = torch.empty((B, int(k)), dtype=torch.int32, device=pts.device)
    with torch.cuda.device(pts.device), prof.kernel("fps", 12.0 * B * n, 8.0 * B * n * k):
        _lib.check(_lib.lib().usip_fps_f32(_ptr(pts), _ptr(first_idx), _ptr(out), B, n, int(k), _stream(pts)),
                   "usip_fps_f32")
    return out


def nms(keypoints: torch.Tensor, sigmas: torch.Tensor, radius: float):
    """f-4: greedy NMS by sigma. keypoints f32 [B,3,M], sigmas f32 [B,M] -> (order i32 [B,M], count i32 [B])."""
    _need_pts(keypoints, "keypoints")
    _need(sigmas, "sigmas", torch.float32)
    B, _, M = keypoints.shape
    order = torch.zeros((B, M), dtype=torch.int32, device=keypoints.device)
    count = torch.empty((B,), dtype=torch.int32, device=keypoints.device)
    with torch.cuda.device(keypoints.device), prof.kernel("nms", 16.0 * B * M):
        _lib.check(_lib.lib().usip_nms_f32(_ptr(keypoints), _ptr(sigmas), float(radius), _ptr(order), _ptr(count),
                                           B, M, _stream(keypoints)), "usip_nms_f32")
    return order, count


def ball_query_cpu(dist: torch.Tensor, radius: float, K: int) -> torch.Tensor:
    """Host twin of ball_query (BASELINE configs[0], the CPU plumbing case): dist f32 [B,M,N] on the HOST -> i32 [B,M,K]."""
    if dist.is_cuda or not dist.is_contiguous() or dist.dtype != torch.float32 or dist.dim() != 3:
        raise RuntimeError("node_to_point_dist must be a contiguous CPU float32 tensor [B,M,N]")
    B, M, N = dist.shape
    out = torch.empty((B, M, int(K)), dtype=torch.int32)
    _lib.check(_lib.lib().usip_ball_query_f32_cpu(_ptr(dist), _ptr(out), float(radius), int(K), B, M, N),
               "usip_ball_query_f32_cpu")
    return out


def pairwise_dist_cpu(a: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """Host twin of pairwise_dist: a [B,3,M], x [B,3,N] on the HOST -> [B,M,N]."""
    for t, name in ((a, "a"), (x, "x")):
        if t.is_cuda or not t.is_contiguous() or t.dtype != torch.float32 or t.dim() != 3 or t.shape[1] != 3:
            raise RuntimeError("%s must be a contiguous CPU float32 tensor [B,3,*]" % name)
    B, _, M = a.shape
    N = x.shape[2]
    out = torch.empty((B, M, N), dtype=torch.float32)
    _lib.check(_lib.lib().usip_pairwise_dist_f32_cpu(_ptr(a), _ptr(x), _ptr(out), B, M, N), "usip_pairwise_dist_f32_cpu")
    return out


def index_max_cpu(data: torch.Tensor, index: torch.Tensor, K: int, num_threads: int = 1) -> torch.Tensor:
    """index_max.forward_cpu / forward_multi_thread_cpu (index_max.cpp:33-112): HOST tensors."""
    for t, name, dt in ((data, "data", torch.float32), (index, "index", torch.int32)):
        if t.is_cuda or not t.is_contiguous() or t.dtype != dt:
            raise RuntimeError("%s must be a contiguous CPU %s tensor" % (name, dt))
    B, C, N = data.shape
    out = torch.empty((B, C, int(K)), dtype=torch.int32)
    _lib.check(_lib.lib().usip_index_max_f32_cpu(_ptr(data), _ptr(index), _ptr(out), B, C, N, int(K),
                                                 int(num_threads)), "usip_index_max_f32_cpu")
    return out
