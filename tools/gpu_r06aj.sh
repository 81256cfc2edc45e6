#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m pytest tests/test_knn_layer_gpu.py -x -q 2>&1 | tail -3
for t in 1 2 4; do
  echo "== tiles per workgroup $t"; USIP_KNN_TPW=$t timeout 200 python tools/knn_layer_bench.py 2>&1 | grep knn_layer_bwd
done
python - <<'PY'
import torch, sys
sys.path.insert(0, '.')
from usip_amd import ops
g = torch.Generator(device="cpu").manual_seed(0)
db = (torch.randn(16, 3, 512, generator=g) * 10).cuda()
idx = ops.knn(db, db, 16)
cnt = torch.stack([torch.bincount(idx[b].reshape(-1).long(), minlength=512) for b in range(16)])
print("in-degree max per cloud", cnt.max(dim=1)[0].tolist(), "mean", float(cnt.float().mean()))
PY
