import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from usip_amd import ops
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
dev="cuda:0"
for (M,K) in [(256,512),(512,256),(64,128)]:
    for nb in (16,32,64):
        At=torch.randn(K,M,device=dev); X=torch.randn(nb,K,512,device=dev)
        us=t(lambda: ops.mlp_gemm(At,X))
        print("M=%d K=%d nb=%d blocks=%d  %.1f us  %.1f TF" % (M,K,nb, nb*4*((M+127)//128), us, 2.0*M*K*512*nb/us/1e6))
