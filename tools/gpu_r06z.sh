#!/bin/bash
# round 6 session z: non-temporal output stores, one store site per variant build (USIP_ST_NT bit), whole step, same box, alternating
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${TAG:-r06z}_ab.txt; rm -f $OUT
for rnd in 1 2 ${ROUNDS3:+3}; do
for v in product ${VARIANTS:-st1 st2 st4 st8 st16 st32 st64}; do
  L=""; [ $v != product ] && L=tools/variants/libusip_hip_$v.so
  USIP_LIB=$L timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-leg --no-fp32-leg --no-n1-probe --no-kernel-timing 2> gpurun_out/r06z_err_$v.txt | python -c "
import sys, json
for ln in sys.stdin:
    ln = ln.strip()
    if ln.startswith('{'):
        d = json.loads(ln); print('$v round $rnd: %.3f ms/step  %.1f clouds/s' % (d['ms_per_step'], d['value']))
" >> $OUT
done; done
cat $OUT
