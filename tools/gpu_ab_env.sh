#!/bin/bash
# same-box A/B of an environment switch on the step:  gpu_ab_env.sh <tag> <VAR> <value A> <value B> [bench args]
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
T=$1; VAR=$2; A=$3; B=$4; shift 4
for rep in 1 2; do
  for v in "$A" "$B"; do
    env $VAR=$v timeout 600 python bench.py --no-cpu-baseline --no-kernel-leg --no-kernel-timing --no-fp32-leg --steps 200 "$@" > gpurun_out/${T}_tmp.json 2>/dev/null
    python -c "import json; r=json.loads(open('gpurun_out/${T}_tmp.json').read().strip().splitlines()[-1]); print('$VAR=$v', round(r['ms_per_step'],4), round(r['value'],1))"
  done
done
