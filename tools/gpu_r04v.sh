#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
T=$1
( time timeout 900 python -m pytest tests/test_modules_gpu.py -m gpu -q -x -k "config3" ) > gpurun_out/${T}_cfg3.log 2>&1; echo "rc=$?" >> gpurun_out/${T}_cfg3.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
timeout 300 python tools/x2_knob_bench.py x2_direct 1 0 4 > gpurun_out/${T}_knob.txt 2>&1
tail -n 12 gpurun_out/${T}_cfg3.log; tail -3 gpurun_out/${T}_bench.err
python - <<PY
import json
r = json.loads(open("gpurun_out/${T}_bench.json").read().strip().splitlines()[-1])
print({k: r[k] for k in ("value", "ms_per_step", "dtype")})
print("fp32:", r.get("fp32_mfma_only")); print("parity:", r.get("parity_check")); print("roofline:", {k: r["roofline"][k] for k in ("kernel", "frac", "avg_us", "launches_per_step")})
for k in r["kernels"][:40]:
    print("%-46s %5.1f calls %8.1f us share %.3f %-7s frac %s" % (k["kernel"][:46], k["calls_per_step"], k["avg_us"], k["share_of_step"], k["bound"], k["frac"]))
PY
grep -v amdgpu.ids gpurun_out/${T}_knob.txt
