"""Practical ceiling of a read+write streaming kernel on this GPU: device-to-device copies and a read-only sum, on a
ring of buffers larger than the 256 MB Infinity Cache."""
import torch
dev = "cuda:0"
def timed(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(it)]
    for s, e in evs:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    t = sorted(s.elapsed_time(e) for s, e in evs)
    return t[len(t) // 2] * 1e3
for mb in (134, 268, 537):
    n = mb * 1000 * 1000 // 4
    src = [torch.randn(n, device=dev) for _ in range(4)]
    dst = [torch.empty(n, device=dev) for _ in range(4)]
    i = [0]
    def cp():
        i[0] = (i[0] + 1) % 4
        dst[i[0]].copy_(src[i[0]])
    def scale():
        i[0] = (i[0] + 1) % 4
        torch.mul(src[i[0]], 2.0, out=dst[i[0]])
    def rd():
        i[0] = (i[0] + 1) % 4
        return src[i[0]].sum()
    def wr():
        i[0] = (i[0] + 1) % 4
        dst[i[0]].fill_(1.0)
    for name, fn, f in (("copy_", cp, 2), ("mul(out=)", scale, 2), ("sum (read)", rd, 1), ("fill (write)", wr, 1)):
        t = timed(fn)
        print("%4d MB  %-12s %7.1f us  %5.2f TB/s (read+write bytes)" % (mb, name, t, f * n * 4 / t / 1e6), flush=True)
