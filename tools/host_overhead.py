"""How long does the HOST take to enqueue one detector step, eagerly (~250 launches) and as a HIP-graph replay
(2 replays + the input copies), next to how long the GPU takes to run it?  With one process per GPU on one host,
the host side is what bounds data-parallel scaling: the N ranks share the host's cores, not the GPUs.

    python tools/host_overhead.py [--steps 30]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usip_amd import synth                                   # noqa: E402
from usip_amd.networks import DetectorOptions                # noqa: E402
from usip_amd.step import DetectorStep, batch_to_device      # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=30)
a = ap.parse_args()
dev = torch.device("cuda", 0)
opt = DetectorOptions(surface_normal_len=4, node_knn_k_1=16)
batch = batch_to_device(synth.make_pair_batch(1234, 8, 16384, 512, 4, "slab"), dev)
for graph in (False, True):
    torch.manual_seed(0)
    st = DetectorStep("ball", opt, dev, with_optimizer=True, graph=graph)
    for _ in range(6):
        st.step(batch)
    b = (st.static_batch(batch) or batch) if graph else batch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    c0 = time.process_time()
    for _ in range(a.steps):
        st.step(b)
    t1 = time.perf_counter()
    c1 = time.process_time()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-13s host enqueue %.3f ms/step (process CPU %.3f ms/step), wall %.3f ms/step" %
          ("graph replay" if graph else "eager", (t1 - t0) / a.steps * 1e3, (c1 - c0) / a.steps * 1e3,
           (t2 - t0) / a.steps * 1e3), flush=True)
    del st
