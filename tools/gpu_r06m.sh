#!/bin/bash
# round 6 evidence session m: rocprof kernel trace + HBM traffic of the step (product build), SQ / TCP counters of the roofline kernel
# (gemm_x2f forward 512 x 512), the driver's bench command
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 bash tools/profile_roofline.sh r06m > gpurun_out/r06m_profile.log 2>&1
timeout 700 bash tools/pmc_any.sh r06m_x2f gemm_x2f tools/x2_one.py > gpurun_out/r06m_pmc_gemm_x2f.txt 2>&1
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06m_bench_driver.json 2> gpurun_out/r06m_bench_driver.err; cp gpurun_out/bench_full_n1.json gpurun_out/r06m_bench_driver_full.json
rm -rf gpurun_out/prof_r06m/*/pmc_* 2>/dev/null
find gpurun_out/prof_r06m gpurun_out/pmc_r06m* -name "*.csv" -size +2M -delete 2>/dev/null
du -sh gpurun_out/prof_r06m gpurun_out/pmc_r06m* 2>/dev/null | tail -4
head -60 gpurun_out/prof_r06m/summary_r06m.txt; cat gpurun_out/r06m_pmc_gemm_x2f.txt | tail -40
head -c 600 gpurun_out/r06m_bench_driver.json; echo
