"""Repeat three training steps from the same seed N times and report every run that differs from the first in any bit
(which parameters): hunts rare races.   python tools/repro_stress.py <model> <N> [pairs points nodes]   (USIP_MATMUL_MODE,
A/B switches; default size: the reproducibility test's 2 pairs x 2048 points, 64 nodes; "8 16384 512" = the bench's)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usip_amd import ops, synth  # noqa: E402
from usip_amd.networks import DetectorOptions  # noqa: E402
from usip_amd.step import DetectorStep, batch_to_device  # noqa: E402

DEV = torch.device("cuda", 0)
ops.set_matmul_mode(os.environ.get("USIP_MATMUL_MODE", "f32x2"))
model, N = sys.argv[1], int(sys.argv[2])
pairs, points, nodes = (int(v) for v in sys.argv[3:6]) if len(sys.argv) >= 6 else (2, 2048, 64)
opt = DetectorOptions(surface_normal_len=4, node_knn_k_1=8 if nodes < 512 else 16)
batch = batch_to_device(synth.make_pair_batch(21, pairs, points, nodes, 4, "sphere" if nodes < 512 else "slab"), DEV)


def run():
    torch.manual_seed(17)
    st = DetectorStep(model, opt, DEV, with_optimizer=True)
    losses = [st.step(batch).detach().clone() for _ in range(3)]
    return losses, st.bucket.flat.clone(), [p.detach().clone() for p in st.bucket.params], [n for n, _ in st.detector.named_parameters()]


ref = run()
nbad = 0
for i in range(N):
    got = run()
    bad = [(n, float((a - b).abs().max())) for n, a, b in zip(ref[3], ref[2], got[2]) if not torch.equal(a, b)]
    if bad or not torch.equal(ref[1], got[1]):
        nbad += 1
        print("run %d differs: %d parameters, first %s" % (i, len(bad), bad[:4]), flush=True)
print("%s: %d of %d runs differ" % (model, nbad, N))
