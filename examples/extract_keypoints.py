#!/usr/bin/env python
"""Keypoints of synthetic clouds with a trained detector, written in the reference's wire format
(evaluation/save_keypoints.py:336-393: per frame a float32 M x 3 row-major .bin of the sigma-ordered NMS
survivors, at most --top of them).

    python examples/extract_keypoints.py --checkpoint /tmp/detector.pth --out /tmp/keypoints"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usip_amd import inference, synth                        # noqa: E402
from usip_amd.networks import DetectorOptions, build_detector  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="ball", choices=["ball", "som"])
    ap.add_argument("--checkpoint", required=True)
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--n", type=int, default=16384)
    ap.add_argument("--m", type=int, default=512)
    ap.add_argument("--nms-radius", type=float, default=2.0)
    ap.add_argument("--top", type=int, default=128)
    ap.add_argument("--out", default="keypoints")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    opt = DetectorOptions(surface_normal_len=4, node_knn_k_1=16)
    detector = build_detector(args.model, opt).to(dev)
    inference.load_detector_state(detector, torch.load(args.checkpoint, map_location=dev))   # 'module.' keys accepted
    rng = np.random.default_rng(7)
    clouds = np.stack([synth.make_cloud(rng, args.n, "slab") for _ in range(args.frames)])
    normals = np.stack([synth.make_normals(rng, args.n, 4) for _ in range(args.frames)])
    pc, sn = torch.from_numpy(clouds).to(dev), torch.from_numpy(normals).to(dev)
    # SOM nodes by farthest-point sampling on the GPU (the reference: numpy in the loader, first index random)
    first = torch.from_numpy(rng.integers(0, args.n, args.frames).astype(np.int32)).to(dev)
    node = inference.sample_nodes(pc, args.m, first)
    keypoints, sigmas = inference.run_model(detector, pc, sn, node)
    frames = inference.select_keypoints(keypoints, sigmas, args.nms_radius, args.top)
    os.makedirs(args.out, exist_ok=True)
    for i, kp in enumerate(frames):
        path = os.path.join(args.out, "%06d.bin" % i)
        inference.write_keypoints_bin(path, kp)
        print("%s  %d keypoints  sigma range [%.3f, %.3f]" % (path, kp.shape[0], float(sigmas[i].min()), float(sigmas[i].max())))


if __name__ == "__main__":
    main()
