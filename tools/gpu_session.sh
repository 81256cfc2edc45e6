#!/bin/bash
# Generic GPU session: tag + list of steps; every step's output lands in gpurun_out/<tag>_<step>.log
#   usage: bash tools/gpu_session.sh <tag> <step> [<step> ...]     steps: pytest pinned x3test x3bench imsweep host bench bench_x3 profile profile_x3
TAG=$1; shift
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
for STEP in "$@"; do
  case $STEP in
    pytest)    timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest.log ;;
    pytestx)   timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest.log ;;
    pytest_x3) USIP_MATMUL_MODE=f32x3 timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_bf16_mode_gpu.py > gpurun_out/${TAG}_pytest_x3.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_x3.log ;;
    pinned)    timeout 600 python -m pytest tests/test_modules_gpu.py -m gpu -q -k "pinned or two_ranks" > gpurun_out/${TAG}_pinned.log 2>&1; echo "rc=$?" >> gpurun_out/${TAG}_pinned.log ;;
    x2test)    timeout 900 python -m pytest tests/test_f32x2_mode_gpu.py -m gpu -q -x > gpurun_out/${TAG}_x2test.log 2>&1; echo "rc=$?" >> gpurun_out/${TAG}_x2test.log ;;
    x3test)    timeout 900 python -m pytest tests/test_f32x3_mode_gpu.py tests/test_bf16_mode_gpu.py -m gpu -q > gpurun_out/${TAG}_x3test.log 2>&1; echo "rc=$?" >> gpurun_out/${TAG}_x3test.log ;;
    x3bench)   timeout 600 python tools/x3_bench.py > gpurun_out/${TAG}_x3bench.txt 2>&1 ;;
    imsweep)   timeout 600 python tools/index_max_sweep.py > gpurun_out/${TAG}_index_max_sweep.txt 2>&1 ;;
    host)      timeout 300 python tools/host_overhead.py > gpurun_out/${TAG}_host_overhead.txt 2>&1 ;;
    bench)     timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err ;;
    benchq)    timeout 900 python bench.py --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err ;;
    bench_x3)  timeout 900 python bench.py --precision f32x3 --no-cpu-baseline > gpurun_out/${TAG}_bench_x3.json 2> gpurun_out/${TAG}_bench_x3.err ;;
    bench_x2)  timeout 900 python bench.py --precision f32x2 --no-cpu-baseline > gpurun_out/${TAG}_bench_x2.json 2> gpurun_out/${TAG}_bench_x2.err ;;
    modes)     timeout 1200 python -m pytest tests/test_modules_gpu.py -m gpu -q -k "pinned or full_size or graph_replay_equals or reproducible or descriptor_step" > gpurun_out/${TAG}_modes.log 2>&1; echo "rc=$?" >> gpurun_out/${TAG}_modes.log ;;
    bench_f32) timeout 900 python bench.py --precision f32 --no-cpu-baseline > gpurun_out/${TAG}_bench_f32.json 2> gpurun_out/${TAG}_bench_f32.err ;;
    profile_f32) timeout 1200 bash tools/profile_roofline.sh ${TAG}_f32 --precision f32 > gpurun_out/${TAG}_profile_f32.log 2>&1 ;;
    bench_som) timeout 900 python bench.py --model som --no-cpu-baseline > gpurun_out/${TAG}_bench_som.json 2> gpurun_out/${TAG}_bench_som.err ;;
    bench_som_x3) timeout 900 python bench.py --model som --precision f32x3 --no-cpu-baseline > gpurun_out/${TAG}_bench_som_x3.json 2> gpurun_out/${TAG}_bench_som_x3.err ;;
    bench_desc) timeout 900 python bench.py --model descriptor > gpurun_out/${TAG}_bench_desc.json 2> gpurun_out/${TAG}_bench_desc.err ;;
    bench_bf16) timeout 900 python bench.py --precision bf16 --no-cpu-baseline > gpurun_out/${TAG}_bench_bf16.json 2> gpurun_out/${TAG}_bench_bf16.err ;;
    bench_cfg1) for pr in f32x2 f32x3 f32 bf16; do timeout 600 python bench.py --model som --points 5000 --nodes 64 --pairs 24 --precision $pr --no-cpu-baseline --no-kernel-leg > gpurun_out/${TAG}_bench_cfg1_${pr}.json 2> gpurun_out/${TAG}_bench_cfg1_${pr}.err; done ;;
    profile_som) timeout 1200 bash tools/profile_roofline.sh ${TAG}_som --model som > gpurun_out/${TAG}_profile_som.log 2>&1 ;;
    profile)   timeout 1200 bash tools/profile_roofline.sh ${TAG} > gpurun_out/${TAG}_profile.log 2>&1 ;;
    profile_x2) timeout 1200 bash tools/profile_roofline.sh ${TAG}_x2 --precision f32x2 > gpurun_out/${TAG}_profile_x2.log 2>&1 ;;
    profile_x3) timeout 1200 bash tools/profile_roofline.sh ${TAG}_x3 --precision f32x3 > gpurun_out/${TAG}_profile_x3.log 2>&1 ;;
    *) echo "unknown step $STEP" ;;
  esac
  echo "== $STEP done ($(date +%T))"
done
for f in gpurun_out/${TAG}_pytest.log gpurun_out/${TAG}_pytest_x3.log gpurun_out/${TAG}_pinned.log gpurun_out/${TAG}_x3test.log gpurun_out/${TAG}_x2test.log gpurun_out/${TAG}_modes.log; do [ -f $f ] && { echo "--- $f"; tail -n 15 $f; }; done
for f in gpurun_out/${TAG}_x3bench.txt gpurun_out/${TAG}_index_max_sweep.txt gpurun_out/${TAG}_host_overhead.txt; do [ -f $f ] && { echo "--- $f"; cat $f; }; done
for f in gpurun_out/${TAG}_bench*.json; do [ -f $f ] && { echo "--- $f"; head -c 700 $f; echo; }; done
exit 0
