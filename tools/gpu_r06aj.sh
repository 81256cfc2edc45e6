#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m pytest tests/test_knn_layer_gpu.py -x -q 2>&1 | tail -3
timeout 200 python tools/knn_layer_bench.py 2>&1 | grep knn_layer
