#!/bin/bash
# round 4 evidence run: kernel trace + HBM traffic of the bench step, PMC of the direct GEMM, the bench lines
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
T=$1
timeout 1200 bash tools/profile_roofline.sh ${T} > gpurun_out/${T}_profile.log 2>&1
echo "== profile done $(date +%T)"
timeout 700 bash tools/pmc_any.sh ${T} gemm_x2d tools/x2_one.py > gpurun_out/${T}_pmc_x2d.txt 2>&1
echo "== pmc done $(date +%T)"
timeout 900 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
timeout 600 python bench.py --precision f32 --no-cpu-baseline --no-kernel-leg > gpurun_out/${T}_bench_f32.json 2>/dev/null
timeout 600 python bench.py --model som --no-cpu-baseline --no-kernel-leg > gpurun_out/${T}_bench_som.json 2>/dev/null
timeout 600 python bench.py --model descriptor > gpurun_out/${T}_bench_desc.json 2>/dev/null
for pr in f32x2 bf16; do timeout 600 python bench.py --model som --points 5000 --nodes 64 --pairs 24 --precision $pr --no-cpu-baseline --no-kernel-leg > gpurun_out/${T}_bench_cfg1_${pr}.json 2>/dev/null; done
echo "== bench done $(date +%T)"
ls gpurun_out/prof_${T}/ | head; tail -5 gpurun_out/${T}_profile.log; cat gpurun_out/${T}_pmc_x2d.txt | head -40
for f in gpurun_out/${T}_bench*.json; do python -c "import json,sys; r=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', round(r['ms_per_step'],3), round(r['value'],1))"; done
