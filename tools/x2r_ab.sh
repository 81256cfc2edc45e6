#!/bin/bash
# same-box A/B of the register-resident f32x2 GEMM (USIP_X2R=0/1): tests, the conv5 microbench, and the whole step
TAG=$1
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_f32x2_mode_gpu.py -m gpu -q -x 2>&1 | tail -3
timeout 300 python tools/x3_bench.py 2>&1 | grep "M= 128 K= 128"
for i in 1 2 3; do
  for v in 0 1; do
    USIP_X2R=$v timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --no-kernel-leg --steps 300 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('X2R=$v', r['ms_per_step'], r['value'])"
  done
done
