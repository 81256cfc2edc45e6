#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_modules_gpu.py tests/test_knn_layer_gpu.py -x -q > gpurun_out/r06al_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r06al_pytest.log
tail -25 gpurun_out/r06al_pytest.log
