#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/test_modules_gpu.py -m gpu -q -x -k "deferred or reproducible or graph_replay_equals or config3_shape_parity or pinned" > gpurun_out/r05c_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r05c_tests.log
tail -4 gpurun_out/r05c_tests.log
for f in 1 0 1 0; do
  USIP_DEFER_WGRAD=$f timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-fp32-leg --no-kernel-leg > gpurun_out/r05c_ab_$f.json 2>> gpurun_out/r05c_ab.err
  python - <<P
import json
d=json.loads(open("gpurun_out/r05c_ab_$f.json").read().strip().splitlines()[-1])
print("defer=$f  ms=%.4f median=%.4f kernels=%d"%(d["ms_per_step"], d["step_ms_rank0"]["median"], d["kernels_total"]))
P
done 2>&1 | tee gpurun_out/r05c_ab.txt
