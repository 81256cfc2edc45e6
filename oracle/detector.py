"""oracle/detector.py -- TEST INFRASTRUCTURE ONLY.

Plain PyTorch-CPU fp32 restatement of the reference's detector training step, written
functionally over a flat {state_dict key: tensor} parameter dictionary.  It is the
checker the HIP path is compared with (tests/, smoke()) and the "port" timed as
bench.py's cpu_baseline.  It is validated against golden vectors produced by importing
the reference itself (tests/golden/make_golden.py); it never runs inside the product.

Every function cites the reference lines it restates (paths relative to /root/reference).
Dense O(N*M) temporaries are deliberate: this file restates, it does not optimise.
"""
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import native

Params = Dict[str, torch.Tensor]
MATMUL = False     # experiment hook (tools/grad_conditioning.py): the 1x1 convolutions through torch.matmul


class DecisionTape:
    """Every DISCRETE decision of one forward, in call order: SOM assignment, index_max / ball / kNN indices, the
    arg-max of every max-pool over K, the on/off mask of every ReLU, the arg-min of every nearest-neighbour
    reduction in the losses.  The step's gradient is a smooth function of the parameters only BETWEEN changes of
    these decisions; values within rounding distance of a tie flip between two correct fp32 evaluations and move
    gradients by O(1e-2) (tools/grad_conditioning.py).  Tests therefore compare gradients at EQUAL decisions:

        tape = DecisionTape(pools=[...], relu_fix=[...])   # record; take the given pool arg-max and the given
        od.TAPE = tape; run fp32; od.TAPE = None            # sparse ReLU overrides (the reference's, from a fixture)
        od.TAPE = DecisionTape(replay=tape.rec); run fp64   # the same decisions again, e.g. in double precision
    """

    def __init__(self, replay=None, pools=None, relu_fix=None):
        self.rec = []                                            # [(kind, tensor)]
        self.replay = list(replay) if replay is not None else None
        self.pools = list(pools) if pools is not None else None
        self.relu_fix = list(relu_fix) if relu_fix is not None else None

    def decide(self, kind, compute):
        if self.replay is not None:
            k, v = self.replay.pop(0)
            assert k == kind, (k, kind)
            self.rec.append((k, v))
            return v
        v = compute()
        if kind == "pool" and self.pools is not None:
            v = self.pools.pop(0).to(v.dtype).reshape(v.shape)
        if kind == "relu" and self.relu_fix is not None:
            idx, on = self.relu_fix.pop(0)
            v = v.clone()
            v.view(-1)[idx] = on
        self.rec.append((kind, v.detach()))
        return v


TAPE = None


def _decide(kind, compute):
    return compute() if TAPE is None else TAPE.decide(kind, compute)


def _min_over(diff: torch.Tensor, dim: int):
    """torch.min(diff, dim) -> (values, arg-min); the arg-min is a decision (autograd routes the gradient to it)."""
    if TAPE is None:
        return torch.min(diff, dim=dim)
    idx = _decide("min", lambda: torch.min(diff, dim=dim)[1])
    return torch.gather(diff, dim, idx.unsqueeze(dim)).squeeze(dim), idx


# --------------------------------------------------------------------------- helpers
def pairwise_norm(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """|a[:, :, i] - b[:, :, j]|_2 -> [B, Ma, Nb], the way every call site of the path
    forms it: torch.norm over an expanded difference (networks.py:694-696,
    layers.py:417-420, losses.py:62-65, losses.py:135-138)."""
    return torch.norm(a.unsqueeze(3) - b.unsqueeze(2), dim=1, keepdim=False)


def index_max_op(data: torch.Tensor, index: torch.Tensor, K: int) -> torch.Tensor:
    """index_max.forward_* (index_max.cpp:73-112) through the C restatement."""
    return _decide("index_max", lambda: torch.from_numpy(
        native.index_max(data.detach().contiguous().numpy(), index.contiguous().numpy(), K)))


def ball_query_op(dist: torch.Tensor, radius: float, K: int) -> torch.Tensor:
    """ball_query.forward_cuda_shared_mem (ball_query_cuda.cu:22-46) through the C restatement."""
    return _decide("ball", lambda: torch.from_numpy(native.ball_query(dist.detach().contiguous().numpy(), float(radius), K)))


def gather_neighbours(x: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """out[b,c,m,k] = x[b,c,idx[b,m,k]] (operations.py:271-287, layers.py:422-426,
    networks.py:699-700)."""
    B, C, _ = x.shape
    _, M, K = idx.shape
    flat = idx.reshape(B, 1, M * K).expand(B, C, M * K)
    return torch.gather(x, 2, flat).view(B, C, M, K)


def shared_mlp(x: torch.Tensor, P: Params, bufs: Optional[Params], prefix: str, train: bool,
               momentum: float = 0.1, eps: float = 1e-5) -> torch.Tensor:
    """One EquivariantLayer / MyConv2d: 1x1 conv (+bias) -> BatchNorm -> ReLU
    (layers.py:208-216, :293-303).  BN and ReLU are present iff the layer owns norm
    parameters (PointNet's last layer and mlp3 do not: layers.py:534-535, networks.py:68)."""
    w = P[prefix + ".conv.weight"]
    b = P[prefix + ".conv.bias"]
    # The same ATen operators the reference's modules call (nn.Conv1d / nn.Conv2d forward):
    # whole-step GRADIENTS are sensitive to rounding-level changes of the forward through
    # max-pool arg-max flips (DESIGN.md "gradient parity"), so the restatement keeps the op
    # choice and is bit-identical to the reference on the pinned platform.
    if MATMUL:                           # experiment hook: the same product through another ATen kernel
        w2 = w.reshape(w.shape[0], w.shape[1])
        y = torch.matmul(w2, x.reshape(x.shape[0], x.shape[1], -1)).reshape((x.shape[0], w.shape[0]) + tuple(x.shape[2:]))
        y = y + b.reshape((1, -1) + (1,) * (x.dim() - 2))
    else:
        y = F.conv2d(x, w, b) if x.dim() == 4 else F.conv1d(x, w, b)
    if prefix + ".norm.weight" in P:
        rm = rv = None
        if bufs is not None:
            rm, rv = bufs[prefix + ".norm.running_mean"], bufs[prefix + ".norm.running_var"]
        if train or rm is None:
            y = F.batch_norm(y, rm, rv, P[prefix + ".norm.weight"], P[prefix + ".norm.bias"],
                             True, momentum, eps)
        else:
            y = F.batch_norm(y, rm, rv, P[prefix + ".norm.weight"], P[prefix + ".norm.bias"],
                             False, momentum, eps)
        if TAPE is None:
            y = torch.relu(y)
        else:                                # the same function with the on/off decision made explicit
            mask = _decide("relu", lambda: (y > 0).detach())
            y = torch.where(mask, y, torch.zeros_like(y))
    return y


# --------------------------------------------------------------------------- SOM front end
def som_assign(node: torch.Tensor, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """som.query_topk with k=1 (util/som.py:31-54): squared distance summed over channels
    0,1,2, nearest node per point.  Returns (min_idx [B,N] int64, count [B,M] int64);
    the reference's dense one-hot `mask` is count's expansion and `mask_row_max` = count>0."""
    diff = x.unsqueeze(3) - node.unsqueeze(2)              # B,3,N,M
    d2 = (diff ** 2).sum(dim=1)                            # B,N,M
    min_idx = _decide("assign", lambda: torch.topk(d2, k=1, dim=2, largest=False, sorted=False)[1].squeeze(2))
    M = node.shape[2]
    count = torch.zeros(x.shape[0], M, dtype=torch.int64).scatter_add_(
        1, min_idx, torch.ones_like(min_idx))
    return min_idx, count


def som_cluster(x: torch.Tensor, min_idx: torch.Tensor, count: torch.Tensor):
    """cluster_mean / centers / x_decentered (networks.py:87-108), all detached there.
    The reference sums x * one_hot over ALL N points per node (dense fp32 sum); we do the
    same dense product so that rounding follows it."""
    M = count.shape[1]
    one_hot = F.one_hot(min_idx, M).to(x.dtype)            # B,N,M
    masked = x.unsqueeze(3) * one_hot.unsqueeze(1)         # B,3,N,M
    cluster_mean = masked.sum(dim=2) / (count.unsqueeze(1).to(x.dtype) + 1e-5)
    centers = (one_hot.unsqueeze(1) * cluster_mean.unsqueeze(2)).sum(dim=3)   # B,3,N
    return cluster_mean.detach(), centers.detach(), (x - centers).detach()


# --------------------------------------------------------------------------- KNN fusion + head
def _max_over_k(h: torch.Tensor, keepdim: bool, pools: Optional[list]):
    """torch.max(h, dim=3) as the reference calls it (networks.py:706,710, layers.py:433,438).  `pools`
    (optional list) receives the arg-max tensor [B,C,M]: the position autograd routes the gradient to."""
    pooled, arg = torch.max(h, dim=3, keepdim=keepdim)
    if TAPE is not None:                 # route through the tape's arg-max (its own when recording without overrides)
        arg = _decide("pool", lambda: arg.detach())
        pooled = torch.gather(h, 3, arg if keepdim else arg.unsqueeze(3))
        pooled = pooled if keepdim else pooled.squeeze(3)
    if pools is not None:
        pools.append(arg.reshape(arg.shape[0], arg.shape[1], arg.shape[2]).detach())
    return pooled


def knn_fusion(P: Params, bufs, prefix: str, query, database, x, K: int, train: bool, pools: Optional[list] = None):
    """GeneralKNNFusionModule.forward (layers.py:401-440). Returns (feature [B,512,M], knn_I)."""
    q = query.detach()
    d = database.detach()
    norm = pairwise_norm(q, d)
    knn_I = _decide("knn", lambda: torch.topk(norm, k=K, dim=2, largest=False, sorted=True)[1])
    coord = gather_neighbours(database, knn_I)
    feat = gather_neighbours(x, knn_I)
    coord = (coord - q.unsqueeze(3)).detach()
    h = torch.cat((coord, feat), dim=1)
    i = 0
    while "%s.layers_before.%d.conv.weight" % (prefix, i) in P:
        h = shared_mlp(h, P, bufs, "%s.layers_before.%d" % (prefix, i), train)
        i += 1
    pooled = _max_over_k(h, True, pools)
    y = torch.cat((pooled.expand_as(h), h), dim=1)
    i = 0
    while "%s.layers_after.%d.conv.weight" % (prefix, i) in P:
        y = shared_mlp(y, P, bufs, "%s.layers_after.%d" % (prefix, i), train)
        i += 1
    out = _max_over_k(y, False, pools)
    return out, knn_I


def head(P: Params, bufs, node_feature, centre, sigma_lower_bound: float, train: bool):
    """mlp1, mlp2, mlp3 + softplus (networks.py:146-154)."""
    y = shared_mlp(node_feature, P, bufs, "mlp1", train)
    y = shared_mlp(y, P, bufs, "mlp2", train)
    ks = shared_mlp(y, P, bufs, "mlp3", train)
    keypoints = ks[:, 0:3, :] + centre
    sigmas = F.softplus(ks[:, 3, :]) + sigma_lower_bound
    return keypoints, sigmas


# --------------------------------------------------------------------------- the two detectors
def rpn_detector_forward(P: Params, bufs, x, sn, node, node_knn_k: int,
                         sigma_lower_bound: float, train: bool = True):
    """RPN_Detector.forward with opt.k == 1 (networks.py:75-162).
    Returns dict(node, keypoints, sigmas, min_idx, first_idx, second_idx, knn_I)."""
    M = node.shape[2]
    min_idx, count = som_assign(node, x)
    has_pts = (count > 0).to(x.dtype).unsqueeze(1)          # mask_row_max
    cluster_mean, _, x_dec = som_cluster(x, min_idx, count)
    h = torch.cat((x_dec, sn), dim=1) if sn is not None and sn.shape[1] > 0 else x_dec
    i = 0
    while "first_pointnet.layers.%d.conv.weight" % i in P:
        h = shared_mlp(h, P, bufs, "first_pointnet.layers.%d" % i, train)
        i += 1
    first = h
    first_idx = index_max_op(first, min_idx.int(), M).long()
    first_max = first.gather(2, first_idx) * has_pts
    scattered = torch.gather(first_max, 2, min_idx.unsqueeze(1).expand(-1, first.shape[1], -1))
    h = torch.cat((first, scattered), dim=1)
    i = 0
    while "second_pointnet.layers.%d.conv.weight" % i in P:
        h = shared_mlp(h, P, bufs, "second_pointnet.layers.%d" % i, train)
        i += 1
    second = h
    second_idx = index_max_op(second, min_idx.int(), M).long()
    second_max = second.gather(2, second_idx) * has_pts
    pools = []
    knn_feat, knn_I = knn_fusion(P, bufs, "knnlayer_1", cluster_mean, cluster_mean, second_max,
                                 node_knn_k, train, pools)
    agg = torch.cat((second_max, knn_feat), dim=1)
    keypoints, sigmas = head(P, bufs, agg, cluster_mean, sigma_lower_bound, train)
    return dict(node=cluster_mean, keypoints=keypoints, sigmas=sigmas, min_idx=min_idx,
                first_idx=first_idx, second_idx=second_idx, knn_I=knn_I, pool_args=pools)


def rpn_detector_ball_forward(P: Params, bufs, x, sn, node, node_knn_k: int,
                              sigma_lower_bound: float, train: bool = True,
                              radius: float = 2, k: int = 64):
    """RPN_Detector_Ball.forward (networks.py:679-738); radius=2, k=64 are hard-coded there
    (:691-692)."""
    x_aug = torch.cat((x, sn), dim=1)
    dist = pairwise_norm(node, x)                           # B,M,N
    ball_idx = ball_query_op(dist, radius, k).long()
    g = gather_neighbours(x_aug, ball_idx)
    g = torch.cat((g[:, 0:3] - node.unsqueeze(3), g[:, 3:]), dim=1)   # in-place at :703
    h = g
    for name in ("conv1", "conv2", "conv3"):
        h = shared_mlp(h, P, bufs, name, train)
    pools = []
    pooled = _max_over_k(h, True, pools)
    h = torch.cat((h, pooled.expand_as(h)), dim=1)          # note order: (features, max) :708
    for name in ("conv4", "conv5"):
        h = shared_mlp(h, P, bufs, name, train)
    second_max = _max_over_k(h, False, pools)
    knn_feat, knn_I = knn_fusion(P, bufs, "knnlayer_1", node, node, second_max, node_knn_k, train, pools)
    agg = torch.cat((second_max, knn_feat), dim=1)
    keypoints, sigmas = head(P, bufs, agg, node, sigma_lower_bound, train)
    return dict(node=node, keypoints=keypoints, sigmas=sigmas, ball_idx=ball_idx, knn_I=knn_I, pool_args=pools)


def knn_rows_canonical(dist: torch.Tensor, k: int) -> torch.Tensor:
    """The k nearest columns of every row of dist [B,M,N], nearest first, ties towards the lower index.
    torch.topk(sorted=False) (networks.py:581) leaves the ORDER of the k picks unspecified and nothing downstream
    depends on it (the neighbours are max-pooled; BatchNorm statistics are sums), so the decision that is pinned
    is the SET, written in this canonical order."""
    order = torch.sort(dist, dim=2, stable=True)[1]
    return order[:, :, :k].contiguous()


def rpn_detector_knn_forward(P: Params, bufs, x, sn, node, node_knn_k: int,
                             sigma_lower_bound: float, train: bool = True, k: int = 64):
    """RPN_Detector_KNN.forward (networks.py:563-608): the Ball detector with the ball query replaced by the
    k = 64 nearest points of every node (k hard-coded at :574, topk(sorted=False) at :581)."""
    x_aug = torch.cat((x, sn), dim=1)
    dist = pairwise_norm(node, x)                           # B,M,N
    nn_idx = _decide("nn", lambda: knn_rows_canonical(dist.detach(), k))
    g = gather_neighbours(x_aug, nn_idx)
    g = torch.cat((g[:, 0:3] - node.unsqueeze(3), g[:, 3:]), dim=1)   # in-place at :587
    h = g
    for name in ("conv1", "conv2", "conv3"):
        h = shared_mlp(h, P, bufs, name, train)
    pools = []
    pooled = _max_over_k(h, True, pools)
    h = torch.cat((h, pooled.expand_as(h)), dim=1)          # (features, max) :593
    for name in ("conv4", "conv5"):
        h = shared_mlp(h, P, bufs, name, train)
    second_max = _max_over_k(h, False, pools)
    knn_feat, knn_I = knn_fusion(P, bufs, "knnlayer_1", node, node, second_max, node_knn_k, train, pools)
    agg = torch.cat((second_max, knn_feat), dim=1)
    keypoints, sigmas = head(P, bufs, agg, node, sigma_lower_bound, train)
    return dict(node=node, keypoints=keypoints, sigmas=sigmas, nn_idx=nn_idx, knn_I=knn_I, pool_args=pools)


# --------------------------------------------------------------------------- losses
def chamfer_prob(src, dst, sigma_src, sigma_dst):
    """ChamferLoss_Brute.forward with both sigmas given (losses.py:59-99).
    Returns (loss, chamfer_pure, chamfer_weighted, src_dst_I, dst_src_I)."""
    diff = pairwise_norm(src, dst)                          # B,M,N
    a, J = _min_over(diff, 2)
    s1 = (sigma_src + torch.gather(sigma_dst, 1, J)) / 2
    fwd = (torch.log(s1) + a / s1).mean()
    c, I = _min_over(diff, 1)
    s2 = (sigma_dst + torch.gather(sigma_src, 1, I)) / 2
    bwd = (torch.log(s2) + c / s2).mean()
    pure = (a.mean() + c.mean()).detach()
    w1 = (1.0 / s1) / torch.mean(1.0 / s1)
    w2 = (1.0 / s2) / torch.mean(1.0 / s2)
    weighted = ((w1 * a).mean() + (w2 * c).mean()).detach()
    return fwd + bwd, pure, weighted, J, I


def chamfer_single_side(kp, pc):
    """SingleSideChamferLoss_Brute.forward (losses.py:125-143) -> [B,M] min distances."""
    d, _ = _min_over(pairwise_norm(kp, pc), 2)
    return d


def point_on_surface(kp, pc, sn):
    """PointOnSurfaceLoss.forward (losses.py:146-187) -> [B,M,1,1]: nearest cloud point p and its normal n (the first
    three channels of sn), (n . (kp - p) / (|kp - p| + 1e-7))^2 through the same ATen calls."""
    B, M = kp.shape[0], kp.shape[2]
    _, I = _min_over(pairwise_norm(kp, pc), 2)
    idx = I.unsqueeze(1).expand(B, 3, M)
    p_sel = torch.gather(pc, 2, idx)
    n_sel = torch.gather(sn, 2, idx)
    diff = kp - p_sel
    unit = diff / (torch.norm(diff, dim=1, keepdim=True) + 1e-7)
    return torch.matmul(n_sel.permute(0, 2, 1).unsqueeze(2), unit.permute(0, 2, 1).unsqueeze(3)) ** 2


# --------------------------------------------------------------------------- the training step
def detector_step(P: Params, bufs, batch: Dict[str, torch.Tensor], model: str, node_knn_k: int,
                  sigma_lower_bound: float, on_pc_alpha: float, on_pc_type: str = "point_to_point"):
    """ModelDetector.optimize minus the optimizer update (keypoint_detector.py:158-205):
    siamese forward on cat(src, dst), rigid transform of the src keypoints, probabilistic
    chamfer + 2x keypoint-on-pc, backward.  P tensors must have requires_grad=True;
    gradients land in P[k].grad.  Returns a dict of every observable of the step."""
    # "lite" (RPN_DetectorLite, networks.py:165-307) is RPN_Detector at half the widths: the same restatement, the
    # widths come with the parameter shapes
    fwd = {"ball": rpn_detector_ball_forward, "knn": rpn_detector_knn_forward,
           "som": rpn_detector_forward, "lite": rpn_detector_forward}[model]
    if "keep_idx" in batch:              # random point dropout (keypoint_detector.py:160-168): ONE index set for all four
        batch = dict(batch)
        idx = batch.pop("keep_idx").long()
        for k in ("src_pc", "src_sn", "dst_pc", "dst_sn"):
            batch[k] = torch.index_select(batch[k], 2, idx)
    B = batch["src_pc"].shape[0]
    out = fwd(P, bufs,
              torch.cat((batch["src_pc"], batch["dst_pc"]), 0),
              torch.cat((batch["src_sn"], batch["dst_sn"]), 0),
              torch.cat((batch["src_node"], batch["dst_node"]), 0),
              node_knn_k, sigma_lower_bound, True)
    kp_src, kp_dst = out["keypoints"][:B], out["keypoints"][B:]
    sg_src, sg_dst = out["sigmas"][:B], out["sigmas"][B:]
    kp_t = torch.matmul(batch["R"], kp_src)                 # :182
    kp_t = kp_t * batch["scale"].unsqueeze(1).unsqueeze(2)  # :183
    kp_t = kp_t + batch["shift"]                            # :184
    loss_chamfer, pure, weighted, _, _ = chamfer_prob(kp_t, kp_dst, sg_src, sg_dst)
    if on_pc_type == "point_to_plane":                      # :197-201
        on_src = point_on_surface(kp_src, batch["src_pc"], batch["src_sn"]).mean() * on_pc_alpha
        on_dst = point_on_surface(kp_dst, batch["dst_pc"], batch["dst_sn"]).mean() * on_pc_alpha
    else:
        on_src = chamfer_single_side(kp_src, batch["src_pc"]).mean() * on_pc_alpha
        on_dst = chamfer_single_side(kp_dst, batch["dst_pc"]).mean() * on_pc_alpha
    loss = loss_chamfer + on_src + on_dst
    loss.backward()
    res = dict(out)
    res.update(loss=loss.detach(), loss_chamfer=loss_chamfer.detach(), chamfer_pure=pure,
               chamfer_weighted=weighted, loss_on_pc_src=on_src.detach(),
               loss_on_pc_dst=on_dst.detach())
    return res


# --------------------------------------------------------------------------- descriptor head (f-1)
def descriptor_forward(P: Params, bufs, x, sn, keypoints, perm, radius: float, K: int, train: bool = True):
    """DescriptorLiteOld.forward (networks.py:333-385) with the random permutation given explicitly."""
    x = x[:, :, perm]
    sn = sn[:, :, perm]
    x_aug = torch.cat((x, sn), dim=1)
    dist = pairwise_norm(keypoints, x).detach()
    ball_idx = ball_query_op(dist, radius, K).long()
    g = gather_neighbours(x_aug, ball_idx)
    g = torch.cat((g[:, 0:3] - keypoints.unsqueeze(3), g[:, 3:]), dim=1)
    h = g
    for name in ("conv1", "conv2", "conv3"):
        h = shared_mlp(h, P, bufs, name, train)
    pools = []
    pooled = _max_over_k(h, True, pools)
    h = torch.cat((h, pooled.expand_as(h)), dim=1)
    h = shared_mlp(h, P, bufs, "conv4", train)
    h = shared_mlp(h, P, bufs, "conv5", train)              # no norm parameters -> plain conv
    d = _max_over_k(h, False, pools)
    d = d / (torch.norm(d, dim=1, keepdim=True) + 1e-5)
    return d, g, ball_idx


def desc_pair_scan_loss(anc, pos, neg, anc_sigmas, gamma: float, sigma_max: float):
    """DescPairScanLoss.forward (losses.py:200-237)."""
    d_pos, J_pos = _min_over(torch.norm(anc.unsqueeze(3) - pos.unsqueeze(2), dim=1), 2)
    d_neg, J_neg = _min_over(torch.norm(anc.unsqueeze(3) - neg.unsqueeze(2), dim=1), 2)
    before = d_pos - d_neg + gamma
    on = _decide("hinge", lambda: (before > 0).detach())     # clamp(min=0): the term is active or not
    active = torch.mean(on.float(), dim=1)
    w = torch.clamp(sigma_max - anc_sigmas, min=0)
    w = (w / torch.mean(w, dim=1, keepdim=True)).detach()
    hinge = torch.clamp(before, min=0) if TAPE is None else torch.where(on, before, torch.zeros_like(before))
    return w * hinge, active, (J_pos, J_neg)


def descriptor_step(P: Params, bufs, batch, perm, radius=2, K=64, gamma=0.5, sigma_max=3.0):
    """ModelDescriptor.optimize minus the optimizer update (keypoint_descriptor.py:126-157)."""
    B = batch["anc_pc"].shape[0]
    desc, feat, ball_idx = descriptor_forward(
        P, bufs, torch.cat((batch["anc_pc"], batch["pos_pc"]), 0), torch.cat((batch["anc_sn"], batch["pos_sn"]), 0),
        torch.cat((batch["anc_kp"], batch["pos_kp"]), 0), perm, radius, K, True)
    anc, pos = desc[:B], desc[B:]
    trip, active, (J_pos, J_neg) = desc_pair_scan_loss(anc, pos, anc[batch["neg_idx"], :, :], batch["anc_sigmas"],
                                                       gamma, sigma_max)
    loss = torch.mean(trip)
    loss.backward()
    return dict(descriptors=desc, x_features=feat, ball_idx=ball_idx, triplet=trip, active=active,
                loss=loss.detach(), nn_pos=J_pos, nn_neg=J_neg)


def to_numpy(d):
    return {k: (v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
            for k, v in d.items() if v is not None and not isinstance(v, list)}
