#!/bin/bash
# same-box A/B of two BUILDS of the library: the working tree's libusip_hip.so against ab/libusip_prev.so (built from
# another commit by hand: git stash; python -m usip_amd.build; cp usip_amd/libusip_hip.so ab/libusip_prev.so; git stash pop)
cd "$GRAFT_REPO_ROOT"; export HSA_ENABLE_IPC_MODE_LEGACY=0
ARGS="--no-cpu-baseline --no-kernel-timing --no-kernel-leg --steps 300 $*"
for i in 1 2 3; do
  for v in prev new; do
    if [ $v = prev ]; then export USIP_LIB=$GRAFT_REPO_ROOT/ab/libusip_prev.so; else unset USIP_LIB; fi
    timeout 300 python bench.py $ARGS 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', r['ms_per_step'], r['value'])"
  done
done
