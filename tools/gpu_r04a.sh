#!/bin/bash
# round 4, session a: L2 -> CU microbenchmark, the direct f32x2 GEMM (gemm_x2d.hip) against the LDS-staged one
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
T=r04a
( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/l2_cu_bw tools/l2_cu_bw.hip 2>/dev/null && timeout 300 /tmp/l2_cu_bw ) > gpurun_out/${T}_l2_cu_bw.txt 2>&1
echo "== l2bw done $(date +%T)"
timeout 900 python -m pytest tests/test_f32x2_mode_gpu.py tests/test_shared_mlp_gpu.py -m gpu -q -x > gpurun_out/${T}_x2test.log 2>&1; echo "rc=$?" >> gpurun_out/${T}_x2test.log
echo "== x2test done $(date +%T)"
timeout 300 python tools/x2_knob_bench.py x2_direct 1 0 > gpurun_out/${T}_knob.txt 2>&1
echo "== knob done $(date +%T)"
timeout 700 bash tools/pmc_any.sh ${T} gemm_x2d tools/x2_one.py > gpurun_out/${T}_pmc_x2d.txt 2>&1
echo "== pmc done $(date +%T)"
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
echo "== bench done $(date +%T)"
tail -n 12 gpurun_out/${T}_x2test.log; cat gpurun_out/${T}_knob.txt; head -c 600 gpurun_out/${T}_bench.json; echo; tail -5 gpurun_out/${T}_bench.err
head -60 gpurun_out/${T}_l2_cu_bw.txt
