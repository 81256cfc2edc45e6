#!/bin/bash
# round 6 session l: full GPU suite on the gemm_x2f build, whole-step A/B (x2_direct 12 = gemm_x2d only / 0 = product), driver's bench command
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r06l_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r06l_pytest.log
tail -4 gpurun_out/r06l_pytest.log
for rnd in 1 2 3; do
for k in 12 0; do
  timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-leg --no-fp32-leg --no-n1-probe --no-kernel-timing --tune x2_direct=$k 2> gpurun_out/r06l_err_$k.txt | python -c "
import sys, json
for ln in sys.stdin:
    ln = ln.strip()
    if ln.startswith('{'):
        d = json.loads(ln); print('x2_direct=$k round $rnd: %.3f ms/step  %.1f clouds/s' % (d['ms_per_step'], d['value']))
" >> gpurun_out/r06l_ab.txt
done; done
cat gpurun_out/r06l_ab.txt
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06l_bench_driver.json 2> gpurun_out/r06l_bench_driver.err; cp gpurun_out/bench_full_n1.json gpurun_out/r06l_bench_driver_full.json
head -c 1500 gpurun_out/r06l_bench_driver.json; echo
