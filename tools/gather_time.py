"""Time usip_group_gather_f32 at the step's two shapes: the Ball front end (7 channels, 512 x 64 positions) and the KNN module
(131 channels from the 512 nodes, 512 x 16 positions).
Round 4 tried two other forms on it: channel chunks (4 x the workgroups: 23.5 vs 23 us) and source rows staged in LDS (18.8 vs
21.9 us for the KNN shape, 27.7 vs 19.2 for the front end's) -- not kept."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usip_amd import ops, _lib
dev = torch.device("cuda:0")
def t(fn, n=20):
    """us per call, 10 calls per replayed graph (the host cannot launch a 10-us kernel back to back)"""
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(10): fn()
    torch.cuda.current_stream().wait_stream(s)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (10 * n)
B = 16
for (C, N, M, K) in ((7, 16384, 512, 64), (131, 512, 512, 16), (128, 512, 512, 16)):
    x = torch.randn(B, C, N, device=dev)
    idx = torch.randint(0, N, (B, M, K), device=dev, dtype=torch.int32)
    out = ops.group_gather(x, idx)
    ref = torch.gather(x.unsqueeze(2).expand(-1, -1, M, -1), 3, idx.long().unsqueeze(1).expand(-1, C, -1, -1))
    print("C=%3d N=%5d M=%d K=%d: %6.1f us  equal to torch.gather: %s" % (C, N, M, K, t(lambda: ops.group_gather(x, idx)), bool(torch.equal(out, ref))))
