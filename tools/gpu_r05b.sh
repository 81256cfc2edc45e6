#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/test_f32x2_mode_gpu.py tests/test_group_gpu.py tests/test_shared_mlp_gpu.py -m gpu -q -x > gpurun_out/r05b_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r05b_tests.log
tail -4 gpurun_out/r05b_tests.log
for f in 0 15 0 1 2 8 4 0 15; do
  timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-fp32-leg --no-kernel-leg --tune r5_forms=$f > gpurun_out/r05b_ab_$f.json 2>> gpurun_out/r05b_ab.err
  python - <<P
import json
d=json.loads(open("gpurun_out/r05b_ab_$f.json").read().strip().splitlines()[-1])
full=json.load(open("gpurun_out/bench_full_n1.json"))
def t(n):
    r=[k for k in full["kernels"] if k["kernel"].startswith(n)]
    return " ".join("%s=%.1f"%(k["kernel"].replace("shared_mlp_",""),k["avg_us"]) for k in r)
print("forms=$f  ms=%.4f median=%.4f | %s | %s | %s"%(d["ms_per_step"], d["step_ms_rank0"]["median"], t("shared_mlp_wgrad 512x512")+" "+t("shared_mlp_wgrad 256x256")+" "+t("shared_mlp_wgrad 512x256"), t("bn_backward_reduce"), t("group_max")))
P
done 2>&1 | tee gpurun_out/r05b_ab.txt
