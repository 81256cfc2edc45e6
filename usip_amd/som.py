"""util/som.py::query_topk of the reference (the only part of that file on the detector path)."""
import torch

from . import ops


def topk_assign(x, node, k: int):
    """The k nearest SOM nodes of every point, in the reference's stacked layout (util/som.py:31-47): -> int64 [B, k*N],
    entry j*N + n = the j-th pick of point n.  k == 1 (every options file of the reference): the HIP nearest-node kernel.
    k > 1 (round 6; `--k`): squared distances summed over the three coordinates in the reference's order (pow, then sum: no
    FMA) and torch.topk(largest=False, sorted=False) on the device -- the ORDER of a point's k picks is unspecified there
    and here; everything the detector computes from them (cluster means, the PointNets over the stacked cloud, index_max
    VALUES) depends on the set only.  Clouds are processed one at a time: the N x M distance matrix of one cloud is 32 MB
    at N = 16384, M = 512 (the reference materialises B x 3 x N x M)."""
    if x.size(1) != 3:
        raise NotImplementedError("usip_amd: SOM assignment expects 3-D coordinates")
    if k == 1:
        return ops.som_assign(x.contiguous(), node.to(x.device).contiguous()).long()
    B, _, N = x.shape
    out = torch.empty((B, k, N), dtype=torch.int64, device=x.device)
    for b in range(B):
        d = ((x[b].unsqueeze(2) - node[b].to(x.device).unsqueeze(1)) ** 2).sum(dim=0)      # [N, M]
        out[b] = torch.topk(d, k=k, dim=1, largest=False, sorted=False)[1].t()
    return out.reshape(B, k * N)


def query_topk(node, x, M, k):
    """node BxCxM, x BxCxN -> (mask B x kN x M int32, mask_row_max B x M int32, min_idx B x kN int64)
    with the reference's layouts (util/som.py:17-54).  The dense one-hot `mask` is materialised here ONLY because
    this drop-in signature returns it -- usip_amd.networks never calls this function and never builds it."""
    min_idx = topk_assign(x, node, int(k))
    mask = torch.nn.functional.one_hot(min_idx, M).int()
    mask_row_max = mask.max(dim=1)[0]
    return mask, mask_row_max, min_idx
