#!/bin/bash
# round-2 GPU session A: parity suite, index_max sweep, host overhead, bench, rocprof passes
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests -m gpu -q -x --deselect tests/test_modules_gpu.py::test_detector_step_gradients_match_reference_with_pinned_routing > gpurun_out/r02a_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r02a_pytest.log
timeout 600 python -m pytest tests/test_modules_gpu.py -m gpu -q -k pinned_routing > gpurun_out/r02a_pinned.log 2>&1; echo "pinned rc=$?" | tee -a gpurun_out/r02a_pinned.log
timeout 300 python tools/index_max_sweep.py > gpurun_out/r02a_index_max_sweep.txt 2>&1
timeout 300 python tools/host_overhead.py > gpurun_out/r02a_host_overhead.txt 2>&1
timeout 900 python bench.py > gpurun_out/r02a_bench.json 2> gpurun_out/r02a_bench.err
timeout 900 bash tools/profile_roofline.sh r02a > gpurun_out/r02a_profile.log 2>&1
tail -5 gpurun_out/r02a_pytest.log gpurun_out/r02a_pinned.log
cat gpurun_out/r02a_index_max_sweep.txt gpurun_out/r02a_host_overhead.txt
head -c 1500 gpurun_out/r02a_bench.json
