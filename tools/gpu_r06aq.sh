#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_modules_gpu.py -x -q -k "first_layer_weight_gradient" 2>&1 | tail -3
OUT=gpurun_out/r06aq_ab.txt; rm -f $OUT
for rnd in 1 2 3; do
for k in 0 1; do
  USIP_WSUM=$k timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-leg --no-fp32-leg --no-n1-probe --no-kernel-timing 2>> gpurun_out/r06aq_err.txt | python -c "
import sys, json
for ln in sys.stdin:
    ln = ln.strip()
    if ln.startswith('{'):
        d = json.loads(ln); print('ball USIP_WSUM=$k round $rnd: %.3f ms/step  %.1f clouds/s' % (d['ms_per_step'], d['value']))
" >> $OUT
done; done
sort $OUT
