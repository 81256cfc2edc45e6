#!/bin/bash
# wgrad_x3 after a change: parity tests, stand-alone timing, per-dispatch HBM fetch inside the step, step time
#   gpu_wgrad_check.sh <tag> [knob value]   (the knob is applied to the stand-alone runs only)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
T=$1; KN=${2:-}; KV=${3:-}
timeout 900 python -m pytest tests/test_f32x2_mode_gpu.py tests/test_shared_mlp_gpu.py -x -q 2>&1 | tail -2
for shape in "512 512" "512 256" "256 256"; do
  python tools/wgrad_one.py $shape 8192 16
  if [ -n "$KN" ]; then python tools/wgrad_one.py $shape 8192 16 $KN $KV; fi
done
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/fetch_$T -o fetch -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-graph --no-fp32-leg > /dev/null 2>&1 )
python - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/fetch_$T/fetch_counter_collection.csv')))
for key in ('wgrad_x3_kernel<2, true, 2','wgrad_x3_kernel<3, true, 2'):
    print(key, [round(2*float(r['Counter_Value'])/1e3,1) for r in rows if key in r['Kernel_Name']][:12])
PY
timeout 300 python bench.py --no-cpu-baseline --no-fp32-leg > gpurun_out/${T}_bench.json
python - <<PY
import json
r=json.loads(open('gpurun_out/${T}_bench.json').read().strip().splitlines()[-1])
print('step ms', r['ms_per_step'])
for k in r['kernels']:
    if 'wgrad' in k['kernel'] and k.get('avg_us',0)>50: print('   %-40s %7.1f us'%(k['kernel'][:40],k['avg_us']))
PY
