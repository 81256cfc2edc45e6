#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_f32x2_mode_gpu.py -m gpu -q -k "full_line or wgrad" > gpurun_out/r05f_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r05f_tests.log; grep -n "FAILED\|passed\|failed" gpurun_out/r05f_tests.log | head -10
bash tools/gpu_r05e.sh
