#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests/test_shared_mlp_gpu.py tests/test_f32x2_mode_gpu.py tests/test_modules_gpu.py -x -q > gpurun_out/r06aa_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r06aa_pytest.log
tail -4 gpurun_out/r06aa_pytest.log
TAG=r06z5 VARIANTS="plain rv1 ldred" ROUNDS3=1 bash tools/gpu_r06z.sh
