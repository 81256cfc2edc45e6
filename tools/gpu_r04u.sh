#!/bin/bash
# round 4: full GPU suite + bench + knob A/B with the direct GEMM
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
T=$1
timeout 300 python tools/x2_knob_bench.py x2_direct 1 0 > gpurun_out/${T}_knob.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/${T}_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/${T}_pytest.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
timeout 600 python bench.py --no-cpu-baseline --no-kernel-leg --no-kernel-timing --steps 200 --tune x2_direct=1 > gpurun_out/${T}_bench_old.json 2> gpurun_out/${T}_bench_old.err
timeout 600 python bench.py --no-cpu-baseline --no-kernel-leg --no-kernel-timing --steps 200 > gpurun_out/${T}_bench_new.json 2> gpurun_out/${T}_bench_new.err
grep -v amdgpu.ids gpurun_out/${T}_knob.txt; tail -n 8 gpurun_out/${T}_pytest.log
for f in bench bench_old bench_new; do python -c "import json,sys; r=json.loads(open('gpurun_out/${T}_$f.json').read().strip().splitlines()[-1]); print('$f', r['ms_per_step'], r['value'], r.get('roofline',{}).get('frac'), r.get('roofline',{}).get('kernel'))"; done
