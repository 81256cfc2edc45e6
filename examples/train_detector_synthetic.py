#!/usr/bin/env python
"""Train the detector for a few steps on synthetic pairs and save a checkpoint whose keys are the reference's
(kitti/train_detector.py saves model.detector.state_dict(); keys = models/networks.py parameter names), so
the file loads into the reference's RPN_Detector(_Ball) and back.

    python examples/train_detector_synthetic.py --model ball --steps 50 --out /tmp/detector.pth
    torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_detector_synthetic.py ...

One process per GPU; every rank owns --pairs pairs per step; gradients are all-reduced over RCCL."""
import argparse
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usip_amd import synth                                   # noqa: E402
from usip_amd.networks import DetectorOptions                # noqa: E402
from usip_amd.step import DetectorStep, batch_to_device      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="ball", choices=["ball", "som"])
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--pairs", type=int, default=8)
    ap.add_argument("--n", type=int, default=16384)
    ap.add_argument("--m", type=int, default=512)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--out", default="detector_synthetic.pth")
    ap.add_argument("--no-graph", action="store_true")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    torch.manual_seed(0)                                     # identical replicas on every rank
    opt = DetectorOptions(surface_normal_len=4, node_knn_k_1=16, lr=args.lr)
    st = DetectorStep(args.model, opt, dev, with_optimizer=True, graph=not args.no_graph and world == 1)
    for it in range(args.steps):
        # a fresh synthetic batch per step, different on every rank (a real loader goes here)
        batch = batch_to_device(synth.make_pair_batch(1000 * it + rank, args.pairs, args.n, args.m, 4, "slab"), dev)
        loss = st.step(batch)
        if rank == 0 and (it % 10 == 0 or it == args.steps - 1):
            print("step %4d  loss %.5f  chamfer %.5f" % (it, float(loss.detach()), float(st.last["chamfer_pure"])))
    if rank == 0:
        torch.save(st.detector.state_dict(), args.out)
        print("saved", args.out)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
