#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
USIP_MATMUL_MODE=f32 timeout 600 python tools/options_grad_table.py > gpurun_out/r06x_table.log 2>&1
tail -120 gpurun_out/r06x_table.log
