#!/bin/bash
# round 4 quick A/B session: f32x2 kernel tests + knob bench   usage: gpu_r04b.sh <tag> [knob values...]
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
T=$1; shift
timeout 900 python -m pytest tests/test_f32x2_mode_gpu.py -m gpu -q -x -k "direct" > gpurun_out/${T}_x2test.log 2>&1; echo "rc=$?" >> gpurun_out/${T}_x2test.log
timeout 300 python tools/x2_knob_bench.py x2_direct "$@" > gpurun_out/${T}_knob.txt 2>&1
tail -n 30 gpurun_out/${T}_x2test.log; cat gpurun_out/${T}_knob.txt
