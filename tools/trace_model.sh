#!/bin/bash
# kernel-trace summary of one model's step:  bash tools/trace_model.sh <tag> <model> [bench flags]  -> gpurun_out/prof_<tag>/
TAG=$1; MODEL=$2; shift 2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o trace -- python $GRAFT_REPO_ROOT/bench.py --model $MODEL --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-graph --no-kernel-leg "$@" > /dev/null 2>&1
