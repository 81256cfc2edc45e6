"""Stand-alone timing of the KNN first-layer kernels at the step's shape (16 clouds, 512 nodes, K = 16, 128 -> 256)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from usip_amd import ops
dev = "cuda:0"
B, C, N, M, K, Cout = 16, 128, 512, 512, 16, 256
P = M * K
def timed(fn, it=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(it)]
    for s, e in evs:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    t = sorted(s.elapsed_time(e) for s, e in evs)
    return t[len(t) // 2] * 1e3
g = torch.Generator(device="cpu").manual_seed(0)
db = (torch.randn(B, 3, N, generator=g) * 10).to(dev)
q = db.clone()
idx = ops.knn(q, db, K)
W = (torch.randn(Cout, 3 + C, generator=g) * 0.1).to(dev)
ring = 3
U = [torch.randn(B, Cout, N, device=dev) for _ in range(ring)]
Y = [torch.randn(B, Cout, P, device=dev) for _ in range(ring)]
dZ = [torch.randn(B, Cout, P, device=dev) for _ in range(ring)]
coef4 = torch.randn(4, Cout, device=dev)
start, perm = ops.csr_by_index(idx.view(B, P), N)
dcoord = ops.group_gather(db, idx, sub=q).view(B, 3, P)
i = [0]
def fwd():
    i[0] = (i[0] + 1) % ring
    return ops.knn_layer_forward(U[i[0]], W, db, q, idx)
def bwd():
    i[0] = (i[0] + 1) % ring
    return ops.knn_layer_backward(dZ[i[0]], Y[i[0]], coef4, True, dcoord, start, perm, M, K)
t = timed(fwd); print("knn_layer_fwd %.1f us  %.2f TB/s (134 MB written)" % (t, 4.0 * B * Cout * P / t / 1e6))
from usip_amd import _lib
for form in (0, 128, 0, 128):                # 0 = two rows per workgroup (the default), 128 = one row
    _lib.lib().usip_set_tuning(b"r5_forms", form)
    t = timed(bwd); print("knn_layer_bwd r5_forms=%d %.1f us  %.2f TB/s (268 MB read)" % (form, t, 8.0 * B * Cout * P / t / 1e6))
_lib.lib().usip_set_tuning(b"r5_forms", 0)
