#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_f32x2_mode_gpu.py -x -q -k "layer_backward or fused_layer" 2>&1 | tail -3
OUT=gpurun_out/r06at_ab.txt; rm -f $OUT
for rnd in 1 2 3; do
for k in 0 256; do
  timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-leg --no-fp32-leg --no-n1-probe --no-kernel-timing --tune r5_forms=$k 2>> gpurun_out/r06at_err.txt | python -c "
import sys, json
for ln in sys.stdin:
    ln = ln.strip()
    if ln.startswith('{'):
        d = json.loads(ln); print('r5_forms=$k round $rnd: %.3f ms/step  %.1f clouds/s' % (d['ms_per_step'], d['value']))
" >> $OUT
done; done
sort $OUT
