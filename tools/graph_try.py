"""Experiment: one detector step (fwd + losses + bwd + Adam) captured in a HIP graph vs eager launches."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usip_amd import synth
from usip_amd.networks import DetectorOptions
from usip_amd.step import DetectorStep, batch_to_device

dev = torch.device("cuda", 0)
torch.manual_seed(0)
opt = DetectorOptions(surface_normal_len=4, node_knn_k_1=16)
st = DetectorStep("ball", opt, dev, with_optimizer=False)
st.optimizer = torch.optim.Adam(st.detector.parameters(), lr=opt.lr, betas=(0.9, 0.999), fused=True, capturable=True)
batch = batch_to_device(synth.make_pair_batch(1234, 8, 16384, 512, 4, "slab"), dev)


def timeit(fn, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        st.step(batch)
torch.cuda.current_stream().wait_stream(s)
print("eager ms/step", timeit(lambda: st.step(batch)))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    loss = st.step(batch)
print("captured; loss", float(loss))
for _ in range(3):
    g.replay()
print("graph ms/step", timeit(g.replay), "loss", float(loss))
print("eager again ms/step", timeit(lambda: st.step(batch)))
