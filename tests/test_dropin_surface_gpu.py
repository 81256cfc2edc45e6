"""INTEGRATION.md levels 1 and 2, EXECUTED: the two detectors run the way a caller of the reference's surface runs them
-- unfused, operator by operator, through public names only -- and must reproduce the reference's fixtures and the
fused `usip_amd.networks` classes.

Public names used (each mirrors a reference name):
    usip_amd.dropin.index_max.forward_cuda_shared_mem / usip_amd.dropin.ball_query.forward_cuda_shared_mem   (level 1)
    usip_amd.som.query_topk, usip_amd.operations.knn_gather_by_indexing,
    usip_amd.layers.MyConv2d / EquivariantLayer / PointNet / GeneralKNNFusionModule                          (level 2)
plus ATen (norm, gather, max, cat, mask products).  The call ORDER is the interface's, as SURVEY.md appendix A writes it
(dense one-hot mask products for the SOM means, index_max -> gather * mask -> broadcast -> cat, distance matrix ->
ball_query -> gather -> decenter -> conv1..3 -> max -> expand + cat -> conv4, conv5 -> max); none of it is taken from the
reference's file, and none of it uses usip_amd.functional, usip_amd.ops or the fused forward() of usip_amd.networks.

What is asserted, per fixture captured from the reference (tests/golden/detector_{som,ball}_micro.npz):
    every index tensor bit-exact; node / keypoints / sigmas within 1e-5 of the REFERENCE's values;
    and the same against the fused detector on the same parameters (the drop-in path and the fast path are one function).
"""
import numpy as np
import pytest
import torch

from conftest import assert_close, load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _detector(fix):
    """The fused detector class, used here as the CONTAINER of the public layer modules (same constructor and
    state_dict keys as the reference's class) and, at the end, as the thing the unfused run must equal."""
    from usip_amd import networks, synth
    from usip_amd.networks import DetectorOptions
    g = load_golden(fix)
    model = str(g["cfg_model"])
    opt = DetectorOptions(surface_normal_len=g["in/src_sn"].shape[1], node_knn_k_1=int(g["cfg_knn"]),
                          loss_sigma_lower_bound=float(g["cfg_sigma_lb"]), keypoint_on_pc_alpha=float(g["cfg_alpha"]))
    net = {"som": networks.RPN_Detector, "ball": networks.RPN_Detector_Ball}[model](opt).to(DEV)
    sd = net.state_dict()
    filled = synth.fill_parameters({k: tuple(v.shape) for k, v in sd.items()})
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)).reshape(sd[k].shape) for k, v in filled.items()})
    net.train()
    t = lambda k: torch.from_numpy(np.ascontiguousarray(g[k])).to(DEV)
    x = torch.cat((t("in/src_pc"), t("in/dst_pc")), 0)
    sn = torch.cat((t("in/src_sn"), t("in/dst_sn")), 0)
    node = torch.cat((t("in/src_node"), t("in/dst_node")), 0)
    return g, opt, net, x, sn, node


def _tail(net, opt, centre, node_feature):
    """knnlayer_1 -> cat -> mlp1 -> mlp2 -> mlp3 -> offsets + centre, softplus + bound (SURVEY appendix A "Head")."""
    knn_feature = net.knnlayer_1(query=centre, database=centre, x=node_feature, K=opt.node_knn_k_1, epoch=None)
    y = net.mlp3(net.mlp2(net.mlp1(torch.cat((node_feature, knn_feature), dim=1))))
    keypoints = y[:, 0:3, :] + centre
    sigmas = torch.nn.functional.softplus(y[:, 3, :]) + opt.loss_sigma_lower_bound
    return keypoints, sigmas


def test_som_detector_through_the_dropin_surface_matches_reference_and_fused():
    from usip_amd import som
    from usip_amd.dropin import index_max
    g, opt, net, x, sn, node = _detector("detector_som_micro.npz")
    B, _, N = x.shape
    M = node.shape[2]
    # F1: query_topk's dense one-hot mask, then the reference's dense products for means / centres
    mask, mask_row_max, min_idx = som.query_topk(node, x, M, k=1)          # [B,N,M] int, [B,M] int, [B,N] int64
    assert mask.shape == (B, N, M) and mask_row_max.shape == (B, M) and min_idx.dtype == torch.int64
    maskf = mask.unsqueeze(1).float()                                       # B,1,N,M
    count = mask.sum(dim=1)                                                 # B,M
    cluster_mean = (x.unsqueeze(3) * maskf).sum(dim=2) / (count.unsqueeze(1).float() + 1e-5)      # B,3,M
    centers = (maskf * cluster_mean.unsqueeze(2)).sum(dim=3)                # B,3,N
    x_dec = (x - centers).detach()
    has_pts = mask_row_max.unsqueeze(1).float()
    # PointNet -> index_max -> gather * mask -> broadcast to the points -> cat -> PointNet -> index_max
    first = net.first_pointnet(torch.cat((x_dec, sn), dim=1), None)
    first_idx = index_max.forward_cuda_shared_mem(first.detach().contiguous(), min_idx.int().contiguous(), M)
    first_max = first.gather(2, first_idx.long()) * has_pts
    scattered = torch.gather(first_max, 2, min_idx.unsqueeze(1).expand(-1, first.shape[1], -1))
    second = net.second_pointnet(torch.cat((first, scattered), dim=1), None)
    second_idx = index_max.forward_cuda_shared_mem(second.detach().contiguous(), min_idx.int().contiguous(), M)
    second_max = second.gather(2, second_idx.long()) * has_pts
    keypoints, sigmas = _tail(net, opt, cluster_mean.detach(), second_max)
    knn_I = net.knnlayer_1.last_knn_I
    # the reference's numbers
    assert np.array_equal(min_idx.cpu().numpy(), g["idx/min_idx"])
    assert np.array_equal(first_idx.cpu().numpy(), g["idx/index_max_0"])
    assert np.array_equal(second_idx.cpu().numpy(), g["idx/index_max_1"])
    assert np.array_equal(knn_I.cpu().numpy(), g["idx/knn_I"])
    assert_close(cluster_mean.cpu().numpy(), g["node"], name="node")
    assert_close(keypoints.detach().cpu().numpy(), g["keypoints"], name="keypoints")
    assert_close(sigmas.detach().cpu().numpy(), g["sigmas"], name="sigmas")
    # ... and the fused detector (coords-in assignment, segment sums, one node per PointNet + pooling) is the same function
    fn, fk, fs, _ = net(x, sn, node, True, None)
    assert torch.equal(net.last_indices["min_idx"].long(), min_idx)
    assert torch.equal(net.last_indices["first_idx"].long(), first_idx.long())
    assert torch.equal(net.last_indices["second_idx"].long(), second_idx.long())
    assert torch.equal(net.last_indices["knn_I"], knn_I)
    assert_close(fn.cpu().numpy(), cluster_mean.cpu().numpy(), name="fused node")
    assert_close(fk.detach().cpu().numpy(), keypoints.detach().cpu().numpy(), name="fused keypoints")
    assert_close(fs.detach().cpu().numpy(), sigmas.detach().cpu().numpy(), name="fused sigmas")
    # gradients flow through the drop-in path (index_max itself has none; gather routes them)
    (keypoints.sum() + sigmas.sum()).backward()
    assert net.first_pointnet.layers[0].conv.weight.grad is not None
    assert float(net.first_pointnet.layers[0].conv.weight.grad.abs().max()) > 0


def test_ball_detector_through_the_dropin_surface_matches_reference_and_fused():
    from usip_amd import operations
    from usip_amd.dropin import ball_query
    g, opt, net, x, sn, node = _detector("detector_ball_micro.npz")
    x_aug = torch.cat((x, sn), dim=1)
    # K4 on the materialised B x M x N distance matrix (the reference's API), then gather + decenter
    dist = torch.norm(node.unsqueeze(3) - x.unsqueeze(2), dim=1, keepdim=False).contiguous()
    ball_idx = ball_query.forward_cuda_shared_mem(dist, 2, 64)              # i32 [B,M,64]
    grouped = operations.knn_gather_by_indexing(x_aug, ball_idx.long())     # B,C,M,K
    grouped = torch.cat((grouped[:, 0:3] - node.unsqueeze(3), grouped[:, 3:]), dim=1)
    h = net.conv3(net.conv2(net.conv1(grouped)))
    pooled, _ = torch.max(h, dim=3, keepdim=True)
    h = torch.cat((h, pooled.expand_as(h)), dim=1)                          # (features, max)
    h = net.conv5(net.conv4(h))
    second_max, _ = torch.max(h, dim=3, keepdim=False)
    keypoints, sigmas = _tail(net, opt, node, second_max)
    knn_I = net.knnlayer_1.last_knn_I
    assert np.array_equal(ball_idx.cpu().numpy(), g["idx/ball_idx"])
    assert np.array_equal(knn_I.cpu().numpy(), g["idx/knn_I"])
    assert_close(keypoints.detach().cpu().numpy(), g["keypoints"], name="keypoints")
    assert_close(sigmas.detach().cpu().numpy(), g["sigmas"], name="sigmas")
    fn, fk, fs, _ = net(x, sn, node, True, None)
    assert torch.equal(net.last_indices["ball_idx"], ball_idx)              # the coords-in kernel takes the same decisions
    assert torch.equal(net.last_indices["knn_I"], knn_I)
    assert_close(fk.detach().cpu().numpy(), keypoints.detach().cpu().numpy(), name="fused keypoints")
    assert_close(fs.detach().cpu().numpy(), sigmas.detach().cpu().numpy(), name="fused sigmas")
    (keypoints.sum() + sigmas.sum()).backward()
    assert float(net.conv1.conv.weight.grad.abs().max()) > 0


def test_knn_fusion_module_unfused_equals_module():
    """GeneralKNNFusionModule written out operator by operator -- norm, topk(sorted), knn_gather_by_indexing, decenter, cat,
    layers_before, max + expand + cat, layers_after, max -- against the module's own (fused) forward."""
    from usip_amd import operations
    g, opt, net, x, sn, node = _detector("detector_ball_micro.npz")
    m = net.knnlayer_1
    B, _, M = node.shape
    torch.manual_seed(3)
    feat = torch.randn(B, net.C1, M, device=DEV)
    K = opt.node_knn_k_1
    norm = torch.norm(node.unsqueeze(3) - node.unsqueeze(2), dim=1)
    knn_I = torch.topk(norm, k=K, dim=2, largest=False, sorted=True)[1]
    coord = (operations.knn_gather_by_indexing(node, knn_I) - node.unsqueeze(3)).detach()
    h = torch.cat((coord, operations.knn_gather_by_indexing(feat, knn_I)), dim=1)
    for layer in m.layers_before:
        h = layer(h)
    pooled, _ = torch.max(h, dim=3, keepdim=True)
    y = torch.cat((pooled.expand_as(h), h), dim=1)                          # (max, features): layers.py order
    for layer in m.layers_after:
        y = layer(y)
    out, _ = torch.max(y, dim=3, keepdim=False)
    fused = m(query=node, database=node, x=feat, K=K, epoch=None)
    assert torch.equal(m.last_knn_I.long(), knn_I)
    assert_close(fused.detach().cpu().numpy(), out.detach().cpu().numpy(), name="knn fusion")
