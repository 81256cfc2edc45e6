"""The detectors of the path, with the reference's class names, constructor, forward
signature and state_dict keys (models/networks.py:20-162 RPN_Detector, :165-307 RPN_DetectorLite,
:482-608 RPN_Detector_KNN, :611-738 RPN_Detector_Ball), written on the fused HIP operators: no
B x N x M temporaries, no dense one-hot masks, distances in the kernels.

forward(x Bx3xN, sn BxCsxN, node Bx3xM, is_train=False, epoch=None)
    -> (nodes Bx3xM, keypoints Bx3xM, sigmas BxM, None)
`is_train` is accepted and ignored as in the reference; train/eval is module state.
"""
import torch
import torch.nn as nn

from . import functional as Fh
from . import ops
from torch.nn.modules.batchnorm import _BatchNorm

from .layers import EquivariantLayer, GeneralKNNFusionModule, MyConv2d, PointNet, pooled_concat_layer


def _bn_kw(opt):
    return dict(momentum=opt.bn_momentum, bn_momentum_decay_step=opt.bn_momentum_decay_step,
                bn_momentum_decay=opt.bn_momentum_decay)


class _DetectorTail(nn.Module):
    """knnlayer_1 + mlp1..3 + softplus, shared by every detector (networks.py:41-72, :135-154)."""

    C2_WIDTH = 512

    def _build_tail(self, opt):
        assert opt.node_knn_k_1 >= 2
        self.C2 = self.C2_WIDTH
        self.knnlayer_1 = GeneralKNNFusionModule(3 + self.C1, (self.C2 // 2, self.C2 // 2, self.C2 // 2),
                                                 (self.C2, self.C2), activation=opt.activation,
                                                 normalization=opt.normalization, **_bn_kw(opt))
        self.mlp1 = EquivariantLayer(self.C1 + self.C2, 512, activation=opt.activation,
                                     normalization=opt.normalization, **_bn_kw(opt))
        self.mlp2 = EquivariantLayer(512, 256, activation=opt.activation,
                                     normalization=opt.normalization, **_bn_kw(opt))
        self.mlp3 = EquivariantLayer(256, 4, activation=None, normalization=None)
        self.mlp3.conv.weight.data.normal_(0, 1e-4)                      # networks.py:70-71
        self.mlp3.conv.bias.data.zero_()
        self.softplus = torch.nn.Softplus()

    def _tail(self, centre, node_feature, epoch):
        knn_feature = self.knnlayer_1(query=centre, database=centre, x=node_feature,
                                      K=self.opt.node_knn_k_1, epoch=epoch)
        agg = torch.cat((node_feature, knn_feature), dim=1)
        y = self.mlp2(self.mlp1(agg, defer=True), defer=True)            # no epoch: networks.py:147-148
        ks = self.mlp3(y)
        # offset + centre, softplus(raw sigma) + lower bound (networks.py:150-154): one launch each way
        return Fh.detector_head(ks, centre, self.opt.loss_sigma_lower_bound)


class RPN_Detector(_DetectorTail):
    """SOM variant: nearest-node assignment -> PointNet -> index_max -> PointNet -> index_max
    -> node KNN fusion -> head (models/networks.py:20-162)."""

    C1_WIDTH = 128

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.C1 = self.C1_WIDTH
        h = self.C1 // 2
        self.first_pointnet = PointNet(3 + opt.surface_normal_len, [h, h, h], activation=opt.activation,
                                       normalization=opt.normalization, **_bn_kw(opt))
        self.second_pointnet = PointNet(self.C1, [self.C1, self.C1], activation=opt.activation,
                                        normalization=opt.normalization, **_bn_kw(opt))
        self._build_tail(opt)

    def forward(self, x, sn, node, is_train=False, epoch=None):
        Fh.require_device(x, "RPN_Detector")
        B, _, N = x.shape
        M = node.shape[2]
        x = x.contiguous()
        k = int(getattr(self.opt, "k", 1))
        if k == 1:
            min_idx32 = ops.som_assign(x, node.contiguous())              # som.py:31-39
        else:
            # --k > 1 (networks.py:85-92): every point is assigned to its k nearest nodes and the cloud is stacked k times;
            # from here on that is the k = 1 computation over k*N points with a GIVEN assignment
            from . import som
            min_idx32 = som.topk_assign(x, node.contiguous(), k).int().contiguous()
            x = x.repeat(1, 1, k).contiguous()
            sn = sn.repeat(1, 1, k)
            N = k * N
        # the assignment sorted by node, once: cluster sums, and every "sum over a node's points" of the backward
        csr = ops.csr_by_index(min_idx32, M) if (Fh.SEGMENT_BACKWARD and ops.segment_sum_supported(M, N)) else None
        cluster_mean, count, x_dec = ops.som_cluster(x, min_idx32, M, csr=csr)   # networks.py:87-107
        self.last_indices = dict(min_idx=min_idx32)
        feat_in = torch.cat((x_dec, sn), dim=1) if self.opt.surface_normal_len >= 1 else x_dec
        if csr is not None:
            # PointNet -> index_max -> gather * mask -> broadcast -> cat as one node per PointNet (networks.py:114-133)
            both, first_idx = self.first_pointnet.forward_som_pooled(feat_in, epoch, min_idx32, count, csr, M, True)
            second_max, second_idx = self.second_pointnet.forward_som_pooled(both, epoch, min_idx32, count, csr, M,
                                                                             False)
            self.last_indices.update(first_idx=first_idx, second_idx=second_idx)     # int32 (the kernels' own)
            keypoints, sigmas = self._tail(cluster_mean, second_max, epoch)
            self.last_indices["knn_I"] = self.knnlayer_1.last_knn_I
            return cluster_mean, keypoints, sigmas, None
        has_pts = (count > 0).to(x.dtype).unsqueeze(1)                    # mask_row_max
        first = self.first_pointnet(feat_in, epoch)
        first_idx = ops.index_max(first.detach().contiguous(), min_idx32, M).long()   # networks.py:117-118
        first_max = first.gather(2, first_idx) * has_pts
        scattered = Fh.cluster_broadcast(first_max, min_idx32)             # networks.py:119-125
        second = self.second_pointnet(torch.cat((first, scattered), dim=1), epoch)
        second_idx = ops.index_max(second.detach().contiguous(), min_idx32, M).long()  # networks.py:130-131
        second_max = second.gather(2, second_idx) * has_pts
        self.last_indices.update(first_idx=first_idx, second_idx=second_idx)
        keypoints, sigmas = self._tail(cluster_mean, second_max, epoch)
        self.last_indices["knn_I"] = self.knnlayer_1.last_knn_I
        return cluster_mean, keypoints, sigmas, None


class RPN_DetectorLite(RPN_Detector):
    """RPN_Detector at half the widths (C1 = 64, C2 = 256; models/networks.py:165-307) -- what every indoor
    train_detector.py of the reference builds (match3d, scenenn: models/keypoint_detector.py:18-19).  Same call
    sequence, same state_dict keys; the kernels pick their own tile shapes for the narrower layers."""
    C1_WIDTH = 64
    C2_WIDTH = 256


class RPN_Detector_Ball(_DetectorTail):
    """Ball-query variant (radius 2, 64 samples, both hard-coded in the reference:
    models/networks.py:691-692): ball grouping -> grouped shared MLP -> max -> concat -> MLP ->
    max -> node KNN fusion -> head (networks.py:611-738)."""

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.C1 = 128
        h = self.C1 // 2
        kw = dict(kernel_size=(1, 1), stride=1, padding=0, bias=True, activation=opt.activation,
                  normalization=opt.normalization, **_bn_kw(opt))
        self.conv1 = MyConv2d(3 + opt.surface_normal_len, h, **kw)
        self.conv2 = MyConv2d(h, h, **kw)
        self.conv3 = MyConv2d(h, h, **kw)
        self.conv4 = MyConv2d(self.C1, self.C1, **kw)
        self.conv5 = MyConv2d(self.C1, self.C1, **kw)
        self._build_tail(opt)
        self.ball_radius = 2
        self.ball_k = 64

    def _neighbourhoods(self, node, x):
        """-> (int32 [B,M,K] point indices of every node's neighbourhood, name of the index tensor)."""
        return ops.ball_query_coords(node, x, self.ball_radius, self.ball_k), "ball_idx"   # :694-698 fused

    def forward(self, x, sn, node, is_train=False, epoch=None):
        Fh.require_device(x, type(self).__name__)
        x = x.contiguous()
        node = node.contiguous()
        x_aug = torch.cat((x, sn), dim=1)
        ball_idx32, idx_name = self._neighbourhoods(node, x)
        g = ops.group_gather(x_aug, ball_idx32, sub=node)                 # gather + decenter :699-703
        # activations stay lazy between the layers: BN+ReLU is applied by the consumer's prologue
        h = self.conv3(self.conv2(self.conv1(g, defer=True), defer=True), defer=True)   # no epoch: networks.py:705
        pooled, h = Fh.group_max_fork(h)                                  # :706
        h = pooled_concat_layer(self.conv4, h, pooled, False)             # cat(h, expand(max)) :708-709 (no epoch: :709)
        if isinstance(getattr(self.conv5, "norm", None), _BatchNorm) and self.conv5.activation == "relu":
            second_max = Fh.conv1x1_bn_relu_max(h, self.conv5.conv.weight, self.conv5.conv.bias,
                                                self.conv5.norm)          # conv5 + max over K fused :709-710
        else:
            second_max = Fh.group_max(self.conv5(h, defer=True))
        keypoints, sigmas = self._tail(node, second_max, epoch)
        self.last_indices = {idx_name: ball_idx32, "knn_I": self.knnlayer_1.last_knn_I}   # int32 (the kernels' own)
        return node, keypoints, sigmas, None


class RPN_Detector_KNN(RPN_Detector_Ball):
    """RPN_Detector_Ball with the ball query replaced by the k = 64 nearest points of every node
    (models/networks.py:482-608; k hard-coded at :574, torch.topk(sorted=False) at :581).  Same layers, same
    state_dict keys.  topk(sorted=False) leaves the order of the k picks unspecified and nothing downstream
    depends on it (max over the neighbours, BatchNorm sums), so the neighbourhoods come out nearest first, ties
    towards the lower index -- `last_indices["nn_idx"]`."""

    def _neighbourhoods(self, node, x):
        return ops.knn_points(node, x, self.ball_k), "nn_idx"                              # :576-581 fused


class DescriptorLiteOld(nn.Module):
    """Descriptor head (models/networks.py:310-385): ball grouping around the detected
    keypoints (radius opt.ball_radius, opt.ball_nsamples samples) -> conv1..3 -> max over K -> conv4 on
    cat(features, max) -> conv5 (plain conv) -> max over K -> L2 normalisation.
    forward(x Bx3xN, sn BxCsxN, keypoints Bx3xM) -> (descriptor BxCxM, x_features Bx(3+Cs)xMxK).

    The reference permutes the points with np.random.permutation on every call (networks.py:345-347)
    because ball_query keeps the FIRST K points inside the ball; set `fixed_permutation` (int64 array or
    tensor of length N) to make a call reproducible."""

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        d = int(opt.descriptor_len)
        kw = dict(kernel_size=(1, 1), stride=1, padding=0, bias=True, activation=opt.activation,
                  normalization=opt.normalization, **_bn_kw(opt))
        self.conv1 = MyConv2d(3 + opt.surface_normal_len, d // 4, **kw)
        self.conv2 = MyConv2d(d // 4, d // 2, **kw)
        self.conv3 = MyConv2d(d // 2, d, **kw)
        self.conv4 = MyConv2d(2 * d, d, **kw)
        self.conv5 = MyConv2d(d, d, kernel_size=(1, 1), stride=1, padding=0, bias=True,
                              activation=None, normalization=None)
        self.fixed_permutation = None

    def forward(self, x, sn, keypoints, is_train=False, epoch=None, perm=None):
        """perm (optional, int64 [N], any device): the point permutation of networks.py:345-347; default: the
        module's fixed_permutation (tests) or a fresh numpy permutation as in the reference."""
        Fh.require_device(x, "DescriptorLiteOld")
        import numpy as np
        N, K = x.shape[2], int(self.opt.ball_nsamples)
        if perm is None:
            perm = self.fixed_permutation if self.fixed_permutation is not None else np.random.permutation(N)
        perm = torch.as_tensor(perm, dtype=torch.int64, device=x.device)       # networks.py:345-347
        x = x[:, :, perm].contiguous()
        x_aug = torch.cat((x, sn[:, :, perm]), dim=1) if self.opt.surface_normal_len > 0 else x
        keypoints = keypoints.detach().contiguous()
        ball_idx32 = ops.ball_query_coords(keypoints, x, float(self.opt.ball_radius), K)      # :352-356 fused
        x_features = ops.group_gather(x_aug.contiguous(), ball_idx32, sub=keypoints)          # :358-370
        h = self.conv3(self.conv2(self.conv1(x_features, defer=True), defer=True), defer=True)   # :373
        pooled, h = Fh.group_max_fork(h)                                                       # :374
        h = pooled_concat_layer(self.conv4, h, pooled, False)                                  # :375-377
        y = self.conv5(h)                                                                      # plain conv
        descriptor = Fh.group_max(y)                                                           # :379
        descriptor = descriptor / (torch.norm(descriptor, dim=1, keepdim=True) + 1e-5)         # :380
        self.last_indices = dict(ball_idx=ball_idx32)
        return descriptor, x_features


class DetectorOptions:
    """The fields of the reference's argparse namespace that the detector path reads
    (kitti/options_detector.py:14-60), with the KITTI defaults."""

    def __init__(self, **kw):
        self.surface_normal_len = 4
        self.activation = "relu"
        self.normalization = "batch"
        self.bn_momentum = 0.1
        self.bn_momentum_decay_step = None
        self.bn_momentum_decay = 0.6
        self.k = 1
        self.node_knn_k_1 = 16
        self.loss_sigma_lower_bound = 0.001
        self.keypoint_on_pc_alpha = 0.01
        self.keypoint_on_pc_type = "point_to_point"
        self.random_pc_dropout_lower_limit = 1.0          # kitti/options_detector.py: off by default
        self.input_pc_num = 16384
        self.lr = 0.001
        # descriptor head (kitti/options_descriptor.py:53-59)
        self.descriptor_len = 128
        self.ball_radius = 2
        self.ball_nsamples = 64
        self.triple_loss_gamma = 0.5
        self.sigma_max = 3.0
        for k, v in kw.items():
            setattr(self, k, v)


DETECTORS = {"som": RPN_Detector, "ball": RPN_Detector_Ball, "lite": RPN_DetectorLite, "knn": RPN_Detector_KNN}


def build_detector(model: str, opt) -> nn.Module:
    return DETECTORS[model](opt)


def detector_param_shapes(model: str, surface_normal_len: int):
    """{state_dict key: shape} of a detector, parameters and BN buffers."""
    net = build_detector(model, DetectorOptions(surface_normal_len=surface_normal_len))
    return {k: tuple(v.shape) for k, v in net.state_dict().items()}
