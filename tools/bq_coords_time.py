"""Time the fused coords-in ball query at the bench's shape (16 clouds, 512 nodes, 16384 points, K = 64, radius 2)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usip_amd import ops, synth
dev = torch.device("cuda:0")
b = synth.make_pair_batch(1234, 8, 16384, 512, 4, "slab")
x = torch.cat((torch.from_numpy(b["src_pc"]), torch.from_numpy(b["dst_pc"]))).to(dev)
node = torch.cat((torch.from_numpy(b["src_node"]), torch.from_numpy(b["dst_node"]))).to(dev)
for _ in range(5):
    idx = ops.ball_query_coords(node, x, 2, 64)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    idx = ops.ball_query_coords(node, x, 2, 64)
e1.record(); torch.cuda.synchronize()
print("ball_query_coords: %.1f us, checksum %d" % (e0.elapsed_time(e1) * 20, int(idx.long().sum())))
