#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r06r_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r06r_pytest.log
tail -25 gpurun_out/r06r_pytest.log
cat gpurun_out/r06_rel_by_channel_*.json
