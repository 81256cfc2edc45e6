"""How long does the host take to ENQUEUE one step vs how long the GPU takes to run it?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usip_amd import synth
from usip_amd.networks import DetectorOptions
from usip_amd.step import DetectorStep, batch_to_device

dev = torch.device("cuda", 0)
opt = DetectorOptions(surface_normal_len=4, node_knn_k_1=16)
torch.manual_seed(0)
st = DetectorStep("ball", opt, dev, with_optimizer=True)
batch = batch_to_device(synth.make_pair_batch(1234, 8, 16384, 512, 4, "slab"), dev)
for _ in range(5):
    st.step(batch)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    st.step(batch)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.2f ms/step, wall %.2f ms/step" % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
