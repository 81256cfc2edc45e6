#!/bin/bash
# round 6, session b: first run of csrc/gemm_x2f.hip -- parity against gemm_x2d.hip, the fp64-truth tests of the family, stand-alone A/B
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_f32x2_mode_gpu.py -x -q -k "one_wave or direct or all_dma or data_gradient" > gpurun_out/r06b_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r06b_pytest.log
tail -15 gpurun_out/r06b_pytest.log
timeout 600 python tools/x2_knob_bench.py x2_direct 0 12 > gpurun_out/r06b_knob.txt 2>&1
cat gpurun_out/r06b_knob.txt
