#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
for v in trace trace_exp3 trace trace_exp3; do
USIP_LIB=tools/variants/libusip_hip_$v.so timeout 300 python tools/x2f_trace.py 512 512 8192 16 > gpurun_out/r06q_${v}_512.txt 2>&1
USIP_LIB=tools/variants/libusip_hip_$v.so timeout 300 python tools/x2f_trace.py 256 256 8192 16 > gpurun_out/r06q_${v}_256.txt 2>&1
echo "== $v"; grep -A3 "^fwd" gpurun_out/r06q_${v}_512.txt gpurun_out/r06q_${v}_256.txt | cut -c1-260
done
