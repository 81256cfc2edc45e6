"""Bit-reproducibility of three training steps with the allocator's free blocks POISONED between the runs (NaN, then
a large finite value): a kernel that reads memory it never wrote (torch.empty) shows up as a difference or a NaN.
  python tools/poison_repro.py [model ...]   (default: knn ball som lite; USIP_MATMUL_MODE selects the arithmetic)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usip_amd import ops, synth  # noqa: E402
from usip_amd.networks import DetectorOptions  # noqa: E402
from usip_amd.step import DetectorStep, batch_to_device  # noqa: E402

DEV = torch.device("cuda", 0)
ops.set_matmul_mode(os.environ.get("USIP_MATMUL_MODE", "f32x2"))


def poison(value):
    blocks = [torch.full((n,), value, dtype=torch.float32, device=DEV) for n in (1 << 28, 1 << 26, 1 << 24, 1 << 22, 1 << 20, 1 << 18, 1 << 16, 1 << 14, 1 << 12, 1 << 10) for _ in range(4)]
    torch.cuda.synchronize()
    del blocks


for model in (sys.argv[1:] or ["knn", "ball", "som", "lite"]):
    opt = DetectorOptions(surface_normal_len=4, node_knn_k_1=8)
    batch = batch_to_device(synth.make_pair_batch(21, 2, 2048, 64, 4, "sphere"), DEV)

    def run():
        torch.manual_seed(17)
        st = DetectorStep(model, opt, DEV, with_optimizer=True)
        losses = [st.step(batch).detach().clone() for _ in range(3)]
        return losses, st.bucket.flat.clone(), [p.detach().clone() for p in st.bucket.params], [n for n, _ in st.detector.named_parameters()]

    ref = run()
    for value in (float("nan"), 3.0e30, -1.0, 0.0):
        poison(value)
        got = run()
        bad = [n for n, a, b in zip(ref[3], ref[2], got[2]) if not torch.equal(a, b)]
        nan = bool(torch.isnan(got[1]).any())
        print("%-5s poison %-8s: losses equal %s, bucket equal %s, NaN in gradients %s, parameters differing %d %s" % (
            model, value, all(torch.equal(a, b) for a, b in zip(ref[0], got[0])), torch.equal(ref[1], got[1]), nan,
            len(bad), bad[:6]), flush=True)
