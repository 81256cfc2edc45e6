#!/bin/bash
# same-box alternating A/B of the whole step: which launches go to csrc/gemm_x2f.hip (knob x2_direct: 12 none, 13 forward only, 14 all but pro 2, 15 all but pro 3, 0 all)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
for rnd in 1 2; do
for k in 12 13 14 15 0; do
  timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-leg --no-fp32-leg --no-n1-probe --no-kernel-timing --tune x2_direct=$k 2> gpurun_out/r06h_err_$k.txt | python -c "
import sys, json
for ln in sys.stdin:
    ln = ln.strip()
    if ln.startswith('{'):
        d = json.loads(ln); print('x2_direct=$k round $rnd: %.3f ms/step  %s clouds/s' % (d['ms_per_step'], d['value']))
" >> gpurun_out/r06h_ab.txt
done; done
cat gpurun_out/r06h_ab.txt; tail -3 gpurun_out/r06h_err_0.txt
