"""Differentiable building blocks of the detector path, on top of the HIP operators (ops.py).

Each function names the reference lines it replaces.  Index-producing steps are not
differentiable in the reference either (it detaches them and routes gradients through
torch.gather); the autograd.Functions here reproduce exactly those gradients.
"""
from typing import Optional, Tuple

import torch
import torch.nn.functional as F

from . import ops, prof


def require_device(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError("usip_amd: %s needs device tensors; the HIP path has no CPU fallback" % what)


# --------------------------------------------------------------------------- distances
class _NearestDistance(torch.autograd.Function):
    """d[b,i] = min_j |a_i - b_j|, J[b,i] = first arg-min; gradients as autograd gives them through
    torch.norm + torch.min in the reference (models/losses.py:62-66, :135-143): the unit vector
    to the selected partner for a, its negative scattered onto b, zero at zero distance."""

    @staticmethod
    def forward(ctx, a, b):
        d, arg = ops.nearest(a.contiguous(), b.contiguous())
        arg = arg.long()
        ctx.save_for_backward(a, b, d, arg)
        ctx.mark_non_differentiable(arg)
        return d, arg

    @staticmethod
    def backward(ctx, gd, _garg):
        a, b, d, arg = ctx.saved_tensors
        sel = torch.gather(b, 2, arg.unsqueeze(1).expand(-1, 3, -1))
        diff = a - sel
        scale = torch.where(d > 0, gd / d, torch.zeros_like(d))          # norm'(0) = 0
        ga = diff * scale.unsqueeze(1)
        gb = None
        if ctx.needs_input_grad[1]:
            gb = torch.zeros_like(b).scatter_add_(2, arg.unsqueeze(1).expand(-1, 3, -1), -ga)
        return (ga if ctx.needs_input_grad[0] else None), gb


def nearest_distance(a: torch.Tensor, b: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """a [B,3,Ma], b [B,3,Nb] -> (min distance [B,Ma], arg-min int64 [B,Ma])."""
    require_device(a, "nearest_distance")
    return _NearestDistance.apply(a, b)


def knn_indices(query: torch.Tensor, database: torch.Tensor, K: int) -> torch.Tensor:
    """The K nearest database points of every query point, nearest first, int64 [B,M,K]
    (torch.norm + topk(sorted=True), models/layers.py:417-421).  The distance matrix comes from
    the HIP kernel in the oracle platform's arithmetic order, so the ordering is the oracle's."""
    require_device(query, "knn_indices")
    dist = ops.pairwise_dist(query.detach().contiguous(), database.detach().contiguous())
    return torch.topk(dist, k=K, dim=2, largest=False, sorted=True)[1]


# --------------------------------------------------------------------------- gathers
def gather_neighbours(x: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """out[b,c,m,k] = x[b,c,idx[b,m,k]]  (models/operations.py:271-287, layers.py:422-426,
    networks.py:699-700); backward = scatter-add, as torch.gather's."""
    B, C, _ = x.shape
    _, M, K = idx.shape
    flat = idx.reshape(B, 1, M * K).expand(B, C, M * K)
    return torch.gather(x, 2, flat).view(B, C, M, K)


# --------------------------------------------------------------------------- shared MLP layer
def conv1x1_bn_act(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor],
                   bn: Optional[torch.nn.modules.batchnorm._BatchNorm], relu: bool) -> torch.Tensor:
    """One shared-MLP layer: 1x1 convolution (+bias) -> BatchNorm (batch statistics when the
    module is in training mode) -> ReLU  (models/layers.py:208-216, :293-303).
    x [B,Cin,*positions], weight [Cout,Cin,1(,1)] -> [B,Cout,*positions]."""
    shape = x.shape
    w2 = weight.reshape(weight.shape[0], weight.shape[1])
    xf = x.reshape(shape[0], shape[1], -1)
    with prof.kernel("shared_mlp_gemm_fwd %dx%d" % (w2.shape[0], w2.shape[1]),
                     flops=2.0 * w2.shape[0] * w2.shape[1] * xf.shape[0] * xf.shape[2]):
        y = torch.matmul(w2, xf)
    if bias is not None:
        y = y + bias.view(1, -1, 1)
    if bn is not None:
        y = F.batch_norm(y, bn.running_mean, bn.running_var, bn.weight, bn.bias,
                         bn.training or bn.running_mean is None, bn.momentum, bn.eps)
    if relu:
        y = torch.relu(y)
    return y.view(shape[0], w2.shape[0], *shape[2:])
