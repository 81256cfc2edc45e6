"""The steps either side of the detector path, on the HIP kernels:

  sample_nodes        farthest-point sampling of SOM nodes on the GPU (the reference does it with numpy in
                      DataLoader workers: data/kitti_detector_loader.py:69-83, :144)
  run_model           eval-mode forward (models/keypoint_detector.py:247-251)
  select_keypoints    sigma-ordered NMS + top-k by sigma (evaluation/save_keypoints.py:180-216, :343-351)
  write_keypoints_bin float32 M x 3 row-major file read by evaluation/matlab/eval_repeatability (:392-393)
  load_detector_state checkpoint loading with the 'module.' prefix fix-up (kitti/train_detector.py:42-51)
"""
from collections import OrderedDict
from typing import List

import numpy as np
import torch

from . import ops


def sample_nodes(pc: torch.Tensor, node_num: int, first_idx: torch.Tensor) -> torch.Tensor:
    """pc f32 [B,3,n] (the loader's random N/3 subset), first_idx i32 [B] -> nodes f32 [B,3,node_num]."""
    idx = ops.fps(pc.contiguous(), first_idx.to(torch.int32).contiguous(), node_num).long()
    return torch.gather(pc, 2, idx.unsqueeze(1).expand(-1, 3, -1))


def run_model(detector: torch.nn.Module, pc, sn, node):
    detector.eval()
    with torch.no_grad():
        _, keypoints, sigmas, _ = detector(pc, sn, node, False, None)
    return keypoints, sigmas


def select_keypoints(keypoints: torch.Tensor, sigmas: torch.Tensor, nms_radius: float,
                     desired_keypoint_num: int) -> List[np.ndarray]:
    """keypoints f32 [B,3,M], sigmas f32 [B,M] -> per cloud a float32 [M',3] array: NMS survivors in ascending
    sigma order, at most desired_keypoint_num of them."""
    B, _, M = keypoints.shape
    if nms_radius < 0.01:                              # save_keypoints.py:188-189: NMS disabled
        order = torch.argsort(sigmas, dim=1, stable=True).int()
        count = torch.full((B,), M, dtype=torch.int32)
    else:
        order, count = ops.nms(keypoints.contiguous(), sigmas.contiguous(), nms_radius)
    order, count = order.cpu().numpy(), count.cpu().numpy()
    kp = keypoints.detach().cpu().numpy()
    return [np.ascontiguousarray(kp[b][:, order[b, :min(int(count[b]), desired_keypoint_num)]].T, dtype=np.float32)
            for b in range(B)]


def write_keypoints_bin(path: str, frame_keypoints: np.ndarray):
    np.asarray(frame_keypoints, dtype=np.float32).tofile(path)


def load_detector_state(detector: torch.nn.Module, state_dict):
    """Accepts checkpoints saved from nn.DataParallel ('module.' prefixed keys) or from a bare module."""
    fixed = OrderedDict((k[len("module."):] if k.startswith("module.") else k, v) for k, v in state_dict.items())
    detector.load_state_dict(fixed)
    return detector
