#!/bin/bash
# round 6 session ap: the first layer's weight gradient from the sums the second layer's fused backward takes on its way (USIP_WSUM=1, default) against its own pass over (dZ, Y) (=0)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r06ap_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r06ap_pytest.log
tail -15 gpurun_out/r06ap_pytest.log
OUT=gpurun_out/r06ap_ab.txt; rm -f $OUT
for rnd in 1 2 3; do
for k in 0 1; do
for model in ball som; do
  USIP_WSUM=$k timeout 300 python bench.py --model $model --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-leg --no-fp32-leg --no-n1-probe --no-kernel-timing 2>> gpurun_out/r06ap_err.txt | python -c "
import sys, json
for ln in sys.stdin:
    ln = ln.strip()
    if ln.startswith('{'):
        d = json.loads(ln); print('$model USIP_WSUM=$k round $rnd: %.3f ms/step  %.1f clouds/s' % (d['ms_per_step'], d['value']))
" >> $OUT
done; done; done
sort $OUT
