"""util/som.py::query_topk of the reference (the only part of that file on the detector path)."""
import torch

from . import ops


def query_topk(node, x, M, k):
    """node BxCxM, x BxCxN -> (mask B x kN x M int32, mask_row_max B x M int32, min_idx B x kN int64)
    with the reference's layouts (util/som.py:17-54).  Only k == 1 is on the path (opt.k is 1 in
    every options file).  The nearest-node search runs in the HIP kernel; the dense one-hot `mask`
    is materialised here ONLY because this drop-in signature returns it -- usip_amd.networks never
    calls this function and never builds it."""
    if k != 1:
        raise NotImplementedError("usip_amd: query_topk is implemented for k == 1 (the detector's value)")
    if x.size(1) != 3:
        raise NotImplementedError("usip_amd: query_topk expects 3-D coordinates")
    min_idx = ops.som_assign(x.contiguous(), node.to(x.device).contiguous()).long()
    mask = torch.nn.functional.one_hot(min_idx, M).int()
    mask_row_max = mask.max(dim=1)[0]
    return mask, mask_row_max, min_idx
