#!/bin/bash
# same-box A/B of an environment switch: bash tools/ab_env2.sh VAR [bench args]   (VAR=0 against VAR=1, three runs each)
VAR=$1; shift
cd "$GRAFT_REPO_ROOT"; export HSA_ENABLE_IPC_MODE_LEGACY=0
for i in 1 2 3; do
  for v in 0 1; do
    env $VAR=$v timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --no-kernel-leg --steps 300 "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', r['ms_per_step'], r['value'])"
  done
done
