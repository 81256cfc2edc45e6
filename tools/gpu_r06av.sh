#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python3 bench.py --model som --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-leg > gpurun_out/r06av_bench_som.json 2> gpurun_out/r06av_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_full_n1.json'))
print(d['ms_per_step'], d['value'])
tot=0
for k in sorted(d['kernels'], key=lambda k:-(k.get('share_of_step') or 0))[:45]:
    print("%-44s calls %4.1f avg %7.1f us share %.4f %s %.0f %s frac %s" % (k['kernel'], k['calls_per_step'], k['avg_us'], k.get('share_of_step') or 0, k.get('bound'), k.get('achieved') or 0, k.get('unit'), k.get('frac')))
PY
