#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
python - > gpurun_out/r06v_narrow_occupancy.txt 2>&1 <<'PY'
import os, sys, runpy
os.environ["USIP_ASSUME_LAUNCH_SAMPLES"] = "1"
sys.path.insert(0, os.getcwd())
import torch
from usip_amd import _lib, ops
dev = "cuda:0"
def timed(fn, it=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(it)]
    for s, e in evs:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    t = sorted(s.elapsed_time(e) for s, e in evs)
    return t[len(t) // 2] * 1e3
nb, K, P = 16, 64, 32768
for M in (64, 128):
    At = torch.randn(K, M, device=dev) * 0.1
    b = torch.randn(M, device=dev)
    coef = torch.stack([1 + 0.1 * torch.randn(K, device=dev), 0.1 * torch.randn(K, device=dev)]).contiguous()
    ring = [torch.randn(nb, K, P, device=dev) for _ in range(3)]
    i = [0]
    for stats in (False, True):
        def f():
            i[0] = (i[0] + 1) % 3
            return ops.mlp_gemm(At, ring[i[0]], b, want_stats=stats, pro=1, coef=coef)
        for rnd in range(2):
            for knob in (0, 64):
                _lib.lib().usip_set_tuning(b"r5_forms", knob)
                t = timed(f)
                print("M=%d stats=%d r5_forms=%d: %.1f us  %.2f TB/s" % (M, stats, knob, t, 4.0 * nb * P * (K + M) / t / 1e6), flush=True)
        _lib.lib().usip_set_tuning(b"r5_forms", 0)
PY
cat gpurun_out/r06v_narrow_occupancy.txt
timeout 700 bash tools/pmc_any.sh r06v_nf narrow_fwd tools/narrow_fwd_probe.py > gpurun_out/r06u_pmc_narrow_fwd.txt 2>&1
find gpurun_out/pmc_r06v* -name "*.csv" -size +2M -delete 2>/dev/null
cat gpurun_out/r06u_pmc_narrow_fwd.txt | cut -c1-150
