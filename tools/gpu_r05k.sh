#!/bin/bash
# round 5 evidence session (final kernels): full GPU suite, the driver's bench command, rocprof trace + HBM traffic of the step, SQ / TCP
# counters of the roofline kernel and of the weight gradient in both forms, the secondary bench lines
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r05k_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05k_pytest.log
tail -3 gpurun_out/r05k_pytest.log
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05k_bench_driver.json 2> gpurun_out/r05k_bench_driver.err; cp gpurun_out/bench_full_n1.json gpurun_out/r05k_bench_driver_full.json
timeout 600 python3 bench.py > gpurun_out/r05k_bench.json 2> gpurun_out/r05k_bench.err; cp gpurun_out/bench_full_n1.json gpurun_out/r05k_bench_full.json
timeout 1500 bash tools/profile_roofline.sh r05k > gpurun_out/r05k_profile.log 2>&1
for w in 0 1; do
  timeout 700 bash tools/pmc_any.sh r05k_wgrad_pool_$w wgrad tools/wgrad_pooled_one.py 512 512 8192 16 r5_forms $w > gpurun_out/r05k_pmc_wgrad_pool_$w.txt 2>&1
  timeout 700 bash tools/pmc_any.sh r05k_wgrad_512x256_$w wgrad tools/wgrad_one.py 512 256 8192 16 r5_forms $w > gpurun_out/r05k_pmc_wgrad_512x256_$w.txt 2>&1
done
timeout 700 bash tools/pmc_any.sh r05k_x2d gemm_x2d tools/x2_one.py > gpurun_out/r05k_pmc_gemm_x2d.txt 2>&1
for m in som descriptor; do timeout 400 python bench.py --model $m --no-cpu-baseline > gpurun_out/r05k_bench_$m.json 2>> gpurun_out/r05k_bench_misc.err; done
timeout 400 python bench.py --precision f32 --no-cpu-baseline --no-kernel-leg > gpurun_out/r05k_bench_f32.json 2>> gpurun_out/r05k_bench_misc.err
for pr in f32x2 bf16; do timeout 400 python bench.py --model som --points 5000 --nodes 64 --pairs 24 --precision $pr --no-cpu-baseline --no-kernel-leg > gpurun_out/r05k_bench_cfg1_$pr.json 2>> gpurun_out/r05k_bench_misc.err; done
rm -rf gpurun_out/prof_r05k/*/pmc_* 2>/dev/null
du -sh gpurun_out/prof_r05k gpurun_out/pmc_r05k* 2>/dev/null | tail -8
find gpurun_out/prof_r05k gpurun_out/pmc_r05k* -name "*.csv" -size +2M -delete 2>/dev/null
head -c 300 gpurun_out/r05k_bench_driver.json; echo; wc -c gpurun_out/r05k_bench*.json
