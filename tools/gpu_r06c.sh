#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
USIP_LIB=tools/variants/libusip_hip_trace.so timeout 300 python tools/x2f_trace.py 512 512 8192 16 > gpurun_out/r06c_trace_512.txt 2>&1
USIP_LIB=tools/variants/libusip_hip_trace.so timeout 300 python tools/x2f_trace.py 256 256 8192 16 > gpurun_out/r06c_trace_256.txt 2>&1
cat gpurun_out/r06c_trace_512.txt gpurun_out/r06c_trace_256.txt | cut -c1-330
