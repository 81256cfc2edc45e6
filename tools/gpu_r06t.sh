#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_f32x2_mode_gpu.py -x -q -k "one_wave or direct or all_dma or data_gradient" > gpurun_out/r06t_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r06t_pytest.log
tail -3 gpurun_out/r06t_pytest.log
timeout 600 python tools/x2_knob_bench.py x2_direct 0 12 > gpurun_out/r06t_knob.txt 2>&1
cat gpurun_out/r06t_knob.txt
USIP_LIB=tools/variants/libusip_hip_trace.so timeout 300 python tools/x2f_trace.py 512 512 8192 16 > gpurun_out/r06t_trace_512.txt 2>&1
USIP_LIB=tools/variants/libusip_hip_trace.so timeout 300 python tools/x2f_trace.py 256 256 8192 16 > gpurun_out/r06t_trace_256.txt 2>&1
grep -A4 "^fwd" gpurun_out/r06t_trace_512.txt gpurun_out/r06t_trace_256.txt | cut -c1-260
