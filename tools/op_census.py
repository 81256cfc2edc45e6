"""ATen operator census of one training step with Python source attribution (torch profiler)."""
import os, sys
import torch
from torch.profiler import profile, ProfilerActivity
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usip_amd import synth
from usip_amd.networks import DetectorOptions
from usip_amd.step import DetectorStep, batch_to_device

dev = torch.device("cuda", 0)
st = DetectorStep("ball", DetectorOptions(surface_normal_len=4, node_knn_k_1=16), dev, with_optimizer=True)
batch = batch_to_device(synth.make_pair_batch(1234, 8, 16384, 512, 4, "slab"), dev)
for _ in range(3):
    st.step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    st.step(batch)
    torch.cuda.synchronize()
names = sys.argv[1:] or ["aten::fill_", "aten::zero_", "aten::copy_", "aten::add"]
for ev in prof.key_averages(group_by_stack_n=6):
    if ev.key in names or any(n.startswith("~") and n[1:] in ev.key for n in names):   # "~sub": substring match
        src = [f for f in ev.stack if "site-packages/torch" not in f and "dist-packages/torch" not in f
               and not f.startswith("<built-in")][:3]
        print("%4d  %-18s %s" % (ev.count, ev.key, " <- ".join(s.split("/")[-1] for s in src) or str(ev.stack[:2])))
