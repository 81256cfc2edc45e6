#!/bin/bash
# nt cache policy (product) against USIP_ST_NT=0 (plain) on the OTHER workloads: SOM model, configs[1] shape, descriptor, fp32-MFMA mode, f32x3
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r06af_ab.txt; rm -f $OUT
run() { # label, lib, args...
  local label=$1 lib=$2; shift 2
  USIP_LIB=$lib timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-leg --no-fp32-leg --no-n1-probe --no-kernel-timing "$@" 2>> gpurun_out/r06af_err.txt | python -c "
import sys, json
for ln in sys.stdin:
    ln = ln.strip()
    if ln.startswith('{'):
        d = json.loads(ln); print('$label: %.3f ms/step  %.1f clouds/s' % (d['ms_per_step'], d['value']))
" >> $OUT
}
for rnd in 1 2; do
  for v in product plain; do
    L=""; [ $v != product ] && L=tools/variants/libusip_hip_$v.so
    run "som $v r$rnd" "$L" --model som
    run "cfg1 $v r$rnd" "$L" --model som --points 5000 --nodes 64 --pairs 24
    run "descriptor $v r$rnd" "$L" --model descriptor
    run "f32 $v r$rnd" "$L" --precision f32
    run "f32x3 $v r$rnd" "$L" --precision f32x3
    run "ball $v r$rnd" "$L"
  done
done
sort $OUT
