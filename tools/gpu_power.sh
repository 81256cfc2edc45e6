#!/bin/bash
# clocks and power while the bench step runs (rocm-smi sampled every 0.5 s)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
rocm-smi --showmaxpower --showpower --showclocks --showperflevel 2>&1 | grep -v "^=\|^$" | head -30
echo "--- during python bench.py --steps 2000 (graph replay)"
( timeout 120 python bench.py --steps 2500 --warmup 10 --no-cpu-baseline --no-kernel-leg --no-kernel-timing --no-fp32-leg > gpurun_out/power_bench.json 2>/dev/null ) &
BP=$!
sleep 45
for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks 2>&1 | grep -i "sclk\|mclk\|power\|fclk" | tr '\n' ' '; echo; sleep 0.7; done
wait $BP
python -c "import json; r=json.loads(open('gpurun_out/power_bench.json').read().strip().splitlines()[-1]); print('step ms', r['ms_per_step'])"
echo "--- during tools/probes/mfma_f16_clock (pure MFMA)"
( for i in 1 2 3 4 5 6 7 8; do timeout 60 tools/probes/mfma_f16_clock > /dev/null; done ) &
BP=$!
sleep 1.0
for i in 1 2 3 4; do rocm-smi --showpower --showclocks 2>&1 | grep -i "sclk\|power" | tr '\n' ' '; echo; sleep 0.3; done
wait $BP
