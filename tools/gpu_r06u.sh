#!/bin/bash
# round 6 session u: SQ / TCP counters of the front-end streaming kernels ("first instrument them")
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 700 bash tools/pmc_any.sh r06u_lb128 layer_bwd_x2 tools/layer_bwd_one.py 128 128 pool > gpurun_out/r06u_pmc_layer_bwd_128_128_pool.txt 2>&1
timeout 700 bash tools/pmc_any.sh r06u_lb64128 layer_bwd_x2 tools/layer_bwd_one.py 64 128 > gpurun_out/r06u_pmc_layer_bwd_64_128.txt 2>&1
timeout 700 bash tools/pmc_any.sh r06u_lb6464 layer_bwd_x2 tools/layer_bwd_one.py 64 64 > gpurun_out/r06u_pmc_layer_bwd_64_64.txt 2>&1
timeout 700 bash tools/pmc_any.sh r06u_nf narrow_fwd tools/narrow_fwd_probe.py > gpurun_out/r06u_pmc_narrow_fwd.txt 2>&1
timeout 700 bash tools/pmc_any.sh r06u_x2r gemm_x2r tools/x2r_one.py > gpurun_out/r06u_pmc_gemm_x2r.txt 2>&1
find gpurun_out/pmc_r06u* -name "*.csv" -size +2M -delete 2>/dev/null
tail -n +1 gpurun_out/r06u_pmc_*.txt | cut -c1-150
