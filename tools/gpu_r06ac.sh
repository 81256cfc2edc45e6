#!/bin/bash
# round 6 evidence session ac: the build with the nt cache policy -- rocprof kernel trace + HBM traffic of the step, PMC of the roofline kernel, the driver's bench command
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 bash tools/profile_roofline.sh r06ac > gpurun_out/r06ac_profile.log 2>&1
timeout 700 bash tools/pmc_any.sh r06ac_x2f gemm_x2f tools/x2_one.py > gpurun_out/r06ac_pmc_gemm_x2f.txt 2>&1
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06ac_bench_driver.json 2> gpurun_out/r06ac_bench_driver.err; cp gpurun_out/bench_full_n1.json gpurun_out/r06ac_bench_driver_full.json
rm -rf gpurun_out/prof_r06ac/*/pmc_* 2>/dev/null
find gpurun_out/prof_r06ac gpurun_out/pmc_r06ac* -name "*.csv" -size +2M -delete 2>/dev/null
head -64 gpurun_out/prof_r06ac/summary_r06ac.txt | cut -c1-200; tail -12 gpurun_out/r06ac_pmc_gemm_x2f.txt | cut -c1-220
head -c 700 gpurun_out/r06ac_bench_driver.json; echo
