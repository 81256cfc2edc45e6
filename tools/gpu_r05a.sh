#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_data_parallel_gpu.py -m gpu -q -x > gpurun_out/r05a_dp.log 2>&1; echo "rc=$?" >> gpurun_out/r05a_dp.log
timeout 900 python -m pytest tests/test_modules_gpu.py tests/test_f32x2_mode_gpu.py -m gpu -q -k "bench or config3_shape_parity or falls_back or dropout or graph" > gpurun_out/r05a_sel.log 2>&1; echo "rc=$?" >> gpurun_out/r05a_sel.log
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05a_bench.json 2> gpurun_out/r05a_bench.err; echo "rc=$?" >> gpurun_out/r05a_bench.err
tail -c 600 gpurun_out/r05a_bench.json | head -c 600; wc -c gpurun_out/r05a_bench.json
tail -5 gpurun_out/r05a_dp.log; tail -5 gpurun_out/r05a_sel.log
