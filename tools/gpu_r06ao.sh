#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_modules_gpu.py -x -q -k "first_layer_weight_gradient" > gpurun_out/r06ao_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r06ao_pytest.log
tail -30 gpurun_out/r06ao_pytest.log
