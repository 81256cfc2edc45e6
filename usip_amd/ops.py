"""Tensor-level wrappers over the C ABI (include/usip_hip.h).

Every function takes contiguous device tensors, allocates its outputs with torch (device
memory + stream plumbing only), enqueues the HIP kernels on torch's CURRENT stream and
returns without synchronising.  Host tensors raise RuntimeError: there is no CPU path here.
"""
import ctypes

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


def som_cluster(x: torch.Tensor, min_idx: torch.Tensor, M: int, decenter: bool = True):
    """a-4: (cluster_mean f32 [B,3,M], count i32 [B,M], x_decentered f32 [B,3,N] or None)."""
    _need_pts(x, "x")
    _need(min_idx, "min_idx", torch.int32)
    B, _, N = x.shape
    mean = torch.empty((B, 3, int(M)), dtype=torch.float32, device=x.device)
    count = torch.empty((B, int(M)), dtype=torch.int32, device=x.device)
    dec = torch.empty_like(x) if decenter else None
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
    with torch.cuda.device(a.device), prof.kernel("nearest", 4.0 * (3 * B * (Ma + Nb) + 2 * B * Ma), 8.0 * B * Ma * Nb):
        _lib.check(_lib.lib().usip_nearest_f32(_ptr(a), _ptr(b), _ptr(d), _ptr(arg), B, Ma, Nb, _stream(a)),
                   "usip_nearest_f32")
    return d, arg


# --------------------------------------------------------------------------- shared MLP
def _opt(t):
    return _ptr(t) if t is not None else None


def mlp_gemm(At: torch.Tensor, X: torch.Tensor, bias=None, want_stats: bool = False, pro: int = 0,
             X2=None, coef=None, tag: str = "fwd"):
    """Y[b] = At^T . pro(X[b]) + bias.  At [K,M] (K-major matrix operand), X [nb,K,P] -> Y [nb,M,P]
    (+ stats [2,tiles,M] when want_stats)."""
    _need(At, "At", torch.float32)
    _need(X, "X", torch.float32)
    K, M = At.shape
    nb, Kx, P = X.shape
    if Kx != K:
        raise RuntimeError("mlp_gemm: At is [%d,%d] but X has %d channels" % (K, M, Kx))
    Y = torch.empty((nb, M, P), dtype=torch.float32, device=X.device)
    stats = None
    if want_stats:
        tiles = _lib.lib().usip_mlp_gemm_tiles(M, P, nb)
        stats = torch.empty((2, tiles, M), dtype=torch.float32, device=X.device)
    with torch.cuda.device(X.device), prof.kernel("shared_mlp_gemm_%s %dx%d" % (tag, M, K),
                                                  4.0 * nb * P * (K * (2 if pro == 2 else 1) + M),
                                                  2.0 * M * K * nb * P):
        _lib.check(_lib.lib().usip_mlp_gemm_f32(_ptr(At), M, _ptr(X), _opt(X2), _opt(coef), int(pro), _opt(bias),
                                                _ptr(Y), _opt(stats), M, K, P, nb, _stream(X)), "usip_mlp_gemm_f32")
    return Y, stats


def bn_finalize(stats, count, gamma, beta, eps, momentum, running_mean, running_var):
    """-> (mean [C], invstd [C], coef [2,C]); updates running_mean/var in place when given."""
    _, tiles, C = stats.shape
    dev = stats.device
    mean = torch.empty(C, dtype=torch.float32, device=dev)
    invstd = torch.empty(C, dtype=torch.float32, device=dev)
    coef = torch.empty((2, C), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev), prof.kernel("bn_finalize", 4.0 * 2 * tiles * C):
        _lib.check(_lib.lib().usip_bn_finalize_f32(_ptr(stats), tiles, C, int(count), _opt(gamma), _opt(beta),
                                                   float(eps), float(momentum), _opt(running_mean),
                                                   _opt(running_var), _ptr(mean), _ptr(invstd), _ptr(coef),
                                                   _stream(stats)), "usip_bn_finalize_f32")
    return mean, invstd, coef


def bn_apply(Y, coef, relu: bool):
    """Z = relu?(Y * coef[0] + coef[1]); Y [nb,C,P]."""
    nb, C, P = Y.shape
    Z = torch.empty_like(Y)
    with torch.cuda.device(Y.device), prof.kernel("bn_apply", 8.0 * nb * C * P):
        _lib.check(_lib.lib().usip_bn_apply_f32(_ptr(Y), _ptr(coef), _ptr(Z), int(bool(relu)), nb, C, P,
                                                _stream(Y)), "usip_bn_apply_f32")
    return Z


def bn_backward_reduce(dZ, Y, coef_fwd, mean, invstd, gamma, relu: bool):
    """-> (dgamma [C], dbeta [C], coef4 [4,C]).  Y None: plain mode, returns (None, sum dZ, None)."""
    nb, C, P = dZ.shape
    dev = dZ.device
    partial = torch.empty(2 * nb * C, dtype=torch.float32, device=dev)
    dbeta = torch.empty(C, dtype=torch.float32, device=dev)
    dgamma = coef4 = None
    if Y is not None:
        dgamma = torch.empty(C, dtype=torch.float32, device=dev)
        coef4 = torch.empty((4, C), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev), prof.kernel("bn_backward_reduce", 4.0 * nb * C * P * (1 if Y is None else 2)):
        _lib.check(_lib.lib().usip_bn_backward_reduce_f32(_ptr(dZ), _opt(Y), _opt(coef_fwd), _opt(mean), _opt(invstd),
                                                          _opt(gamma), int(bool(relu)), _ptr(partial), _opt(dgamma),
                                                          _ptr(dbeta), _opt(coef4), nb, C, P, _stream(dZ)),
                   "usip_bn_backward_reduce_f32")
    return dgamma, dbeta, coef4


def mlp_wgrad(G, X, pro: int = 0, G2=None, coef4=None):
    """dW [M,N] = sum_{b,p} pro(G)[b,m,p] * X[b,n,p]; G [nb,M,P], X [nb,N,P]."""
    nb, M, P = G.shape
    N = X.shape[1]
    dev = G.device
    ws_n = _lib.lib().usip_mlp_wgrad_workspace(M, N, P, nb)
    ws = torch.empty(max(int(ws_n), 1), dtype=torch.float32, device=dev)
    dW = torch.empty((M, N), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev), prof.kernel("shared_mlp_wgrad %dx%d" % (M, N),
                                             4.0 * nb * P * (M * (2 if pro == 2 else 1) + N), 2.0 * M * N * nb * P):
        _lib.check(_lib.lib().usip_mlp_wgrad_f32(_ptr(G), _opt(G2), _opt(coef4), int(pro), _ptr(X), _ptr(ws), _ptr(dW),
                                                 M, N, P, nb, _stream(G)), "usip_mlp_wgrad_f32")
    return dW


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
