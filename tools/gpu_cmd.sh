#!/bin/bash
# run an arbitrary command line on the GPU box with output to gpurun_out/<tag>.txt:  gpu_cmd.sh <tag> <cmd...>
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
T=$1; shift
( eval "$@" ) > gpurun_out/${T}.txt 2>&1
tail -n 60 gpurun_out/${T}.txt
