#!/bin/bash
# round 5, final build: full GPU suite, smoke(), the driver's bench command three times, the default line, secondary lines
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r05h_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05h_pytest.log
tail -3 gpurun_out/r05h_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05h_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r05h_smoke.log; tail -3 gpurun_out/r05h_smoke.log
for i in 1 2 3; do timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05h_bench_driver_$i.json 2> gpurun_out/r05h_bench_driver_$i.err; done
cp gpurun_out/bench_full_n1.json gpurun_out/r05h_bench_driver_full.json
timeout 600 python3 bench.py > gpurun_out/r05h_bench.json 2> gpurun_out/r05h_bench.err; cp gpurun_out/bench_full_n1.json gpurun_out/r05h_bench_full.json
for m in som descriptor; do timeout 400 python bench.py --model $m --no-cpu-baseline > gpurun_out/r05h_bench_$m.json 2>> gpurun_out/r05h_bench_misc.err; done
timeout 400 python bench.py --precision f32 --no-cpu-baseline --no-kernel-leg > gpurun_out/r05h_bench_f32.json 2>> gpurun_out/r05h_bench_misc.err
for pr in f32x2 bf16; do timeout 400 python bench.py --model som --points 5000 --nodes 64 --pairs 24 --precision $pr --no-cpu-baseline --no-kernel-leg > gpurun_out/r05h_bench_cfg1_$pr.json 2>> gpurun_out/r05h_bench_misc.err; done
python - <<P
import json,glob
for f in sorted(glob.glob("gpurun_out/r05h_bench*.json")):
    if "full" in f: continue
    t=open(f).read().strip().splitlines()
    d=json.loads(t[-1]); print(f, len(t[-1]), "%.4f ms %.1f/s"%(d["ms_per_step"], d["value"]), d.get("roofline",{}).get("frac"))
P
