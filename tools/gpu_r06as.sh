#!/bin/bash
# round 6 evidence session as (final build): rocprof kernel trace + HBM traffic of the step, PMC of the roofline kernel, then everything gpu_r06ae.sh runs
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 bash tools/profile_roofline.sh r06as > gpurun_out/r06as_profile.log 2>&1
timeout 700 bash tools/pmc_any.sh r06as_x2f gemm_x2f tools/x2_one.py > gpurun_out/r06as_pmc_gemm_x2f.txt 2>&1
rm -rf gpurun_out/prof_r06as/*/pmc_* 2>/dev/null
find gpurun_out/prof_r06as gpurun_out/pmc_r06as* -name "*.csv" -size +2M -delete 2>/dev/null
head -30 gpurun_out/prof_r06as/summary_r06as.txt | cut -c1-180
TAG=r06as bash tools/gpu_r06ae.sh
