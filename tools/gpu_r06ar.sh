#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
for k in 0 1 0 1; do
USIP_WSUM=$k timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-leg --no-kernel-leg > /dev/null 2>&1
python - <<PY
import json
d=json.load(open('gpurun_out/bench_full_n1.json'))
rows={k['kernel']:k for k in d['kernels']}
out=["WSUM=$k step %.3f" % d['ms_per_step']]
for n in ('shared_mlp_layer_bwd_x2 64x64','shared_mlp_wgrad 64x7','wsum_finalize','bn_backward_finalize'):
    if n in rows: out.append("%s: %.1f us x %.0f" % (n, rows[n]['avg_us'], rows[n]['calls_per_step']))
print(" | ".join(out))
PY
done
