"""ATen operator census of one training step with Python source attribution (torch profiler)."""
import os, sys
from collections import Counter
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
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    st.step(batch)
    torch.cuda.synchronize()
want = ("aten::fill_", "aten::zero_", "aten::copy_", "aten::add", "aten::add_", "aten::contiguous", "aten::clone")
cnt = Counter()
for ev in prof.events():
    if ev.name in want:
        where = "?"
        for fr in ev.stack:
            if "usip_amd" in fr or "bench" in fr or "optim" in fr:
                where = fr.split("/")[-1]
                break
        cnt[(ev.name, where)] += 1
for (name, where), n in cnt.most_common(45):
    print("%4d  %-18s %s" % (n, name, where))
