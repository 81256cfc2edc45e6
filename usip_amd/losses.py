"""Detector losses with the reference's class names and call signatures (models/losses.py),
built on the fused nearest-neighbour kernel instead of materialised B x M x N matrices
(SURVEY 8 a-9, a-10)."""
import torch
import torch.nn as nn

from . import functional as Fh


class ChamferLoss_Brute(nn.Module):
    """Probabilistic (sigma-weighted) chamfer loss (models/losses.py:44-99).
    forward(src Bx3xM, dst Bx3xN, sigma_src BxM, sigma_dst BxN) -> (loss, chamfer_pure, chamfer_weighted)."""

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.dimension = 3

    def forward(self, pc_src_input, pc_dst_input, sigma_src=None, sigma_dst=None):
        a, J = Fh.nearest_distance_i32(pc_src_input, pc_dst_input)      # row minima  (:81)
        c, I = Fh.nearest_distance_i32(pc_dst_input, pc_src_input)      # column minima (:86)
        self.last_indices = (J, I)                                       # int32 (torch.min gives int64)
        if sigma_src is None or sigma_dst is None:                       # losses.py:68-78
            return a + c, a + c, a + c
        return Fh.chamfer_prob(a, J, c, I, sigma_src, sigma_dst)         # :82-99, one launch


class SingleSideChamferLoss_Brute(nn.Module):
    """min_n |src[b,:,m] - dst[b,:,n]| -> BxM (models/losses.py:119-143)."""

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.dimension = 3

    def forward(self, pc_src_input, pc_dst_input):
        return Fh.nearest_distance_i32(pc_src_input, pc_dst_input)[0]


class PointOnSurfaceLoss(nn.Module):
    """Point-to-plane form (models/losses.py:146-187): with p = the cloud point nearest to the keypoint and n its
    normal (the first three channels of sn), (n . (kp - p) / (|kp - p| + 1e-7))^2 -> B x M x 1 x 1.  The arg-min comes
    from the fused nearest-neighbour kernel (no B x M x N matrix); the rest is M-sized ATen arithmetic in the
    reference's order, which autograd differentiates exactly as it does there (the arg-min carries no gradient)."""

    def __init__(self, opt):
        super().__init__()
        self.opt = opt

    def forward(self, keypoint, pc, sn):
        B, M = keypoint.shape[0], keypoint.shape[2]
        _, I = Fh.nearest_distance_i32(keypoint.detach(), pc)
        self.last_indices = I
        idx = I.long().unsqueeze(1).expand(B, 3, M)
        pc_selected = torch.gather(pc, 2, idx)
        sn_selected = torch.gather(sn, 2, idx)                          # channels 0..2 of sn
        diff = keypoint - pc_selected
        unit = diff / (torch.norm(diff, dim=1, keepdim=True) + 1e-7)
        dot = torch.matmul(sn_selected.permute(0, 2, 1).unsqueeze(2), unit.permute(0, 2, 1).unsqueeze(3))
        return dot ** 2                                                 # B x M x 1 x 1


class KeypointOnPCLoss(nn.Module):
    """Keypoint-to-cloud distance (models/losses.py:102-116): point-to-point (sn None: the default of every options
    file) or point-to-plane (sn given; selected by opt.keypoint_on_pc_type, keypoint_detector.py:193-201)."""

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.single_side_chamfer = SingleSideChamferLoss_Brute(opt)
        self.keypoint_on_surface = PointOnSurfaceLoss(opt)

    def forward(self, keypoint, pc, sn=None):
        if sn is not None:
            return self.keypoint_on_surface(keypoint, pc, sn)
        return self.single_side_chamfer(keypoint, pc)


class DescPairScanLoss(nn.Module):
    """Descriptor triplet loss with in-batch negatives (models/losses.py:190-237):
    for every anchor keypoint the nearest positive-scan and nearest negative-scan descriptor,
    clamp(d_pos - d_neg + gamma, 0) weighted by clamp(sigma_max - sigma, 0) normalised to mean 1.
    forward(anc BxCxM, pos BxCxM, neg BxCxM, anc_sigmas BxM) -> (loss BxM, active_percentage B).
    The two B x C x M x M difference tensors of the reference are replaced by the fused nearest kernel."""

    def __init__(self, opt):
        super().__init__()
        self.opt = opt

    def forward(self, anc_descriptors, pos_descriptors, neg_descriptors, anc_sigmas):
        d_pos, j_pos = Fh.nearest_distance_i32(anc_descriptors, pos_descriptors)
        d_neg, j_neg = Fh.nearest_distance_i32(anc_descriptors, neg_descriptors)
        self.last_indices = (j_pos, j_neg)                              # int32 (torch.min gives int64)
        before_clamp = d_pos - d_neg + self.opt.triple_loss_gamma
        active_percentage = torch.mean((before_clamp > 0).float(), dim=1)
        w = torch.clamp(self.opt.sigma_max - anc_sigmas, min=0)
        w = (w / torch.mean(w, dim=1, keepdim=True)).detach()
        return w * torch.clamp(before_clamp, min=0), active_percentage
