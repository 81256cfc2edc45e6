#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_f32x2_mode_gpu.py -x -q -k "one_wave" > gpurun_out/r06k_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r06k_pytest.log
tail -3 gpurun_out/r06k_pytest.log
timeout 600 python tools/x2_knob_bench.py r5_forms 0 64 128 192 256 > gpurun_out/r06k_knob.txt 2>&1
cat gpurun_out/r06k_knob.txt
