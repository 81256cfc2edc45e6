#!/bin/bash
# same-box comparison of tuning-knob settings: bash tools/ab_tune.sh "a=1" "b=2,c=3" ...   ("-" = library defaults)
cd "$GRAFT_REPO_ROOT"; export HSA_ENABLE_IPC_MODE_LEGACY=0
for i in 1 2; do
  for v in "$@"; do
    if [ "$v" = "-" ]; then T=""; else T="--tune $(echo $v | sed 's/,/ --tune /g')"; fi
    timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing --no-kernel-leg --steps 300 $T 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', r['ms_per_step'], r['value'])"
  done
done
