#!/bin/bash
# round 6 evidence session ae (final build): full GPU suite, smoke, the driver's bench command, the default bench, the secondary bench lines
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
T=${TAG:-r06ae}
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/${T}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_pytest.log
tail -4 gpurun_out/${T}_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${T}_smoke.log 2>&1; tail -2 gpurun_out/${T}_smoke.log
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench_driver.json 2> gpurun_out/${T}_bench_driver.err; cp gpurun_out/bench_full_n1.json gpurun_out/${T}_bench_driver_full.json
timeout 600 python3 bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; cp gpurun_out/bench_full_n1.json gpurun_out/${T}_bench_full.json
for m in som descriptor; do timeout 400 python bench.py --model $m --no-cpu-baseline > gpurun_out/${T}_bench_$m.json 2>> gpurun_out/${T}_bench_misc.err; done
timeout 400 python bench.py --precision f32 --no-cpu-baseline --no-kernel-leg > gpurun_out/${T}_bench_f32.json 2>> gpurun_out/${T}_bench_misc.err
timeout 400 python bench.py --precision f32x3 --no-cpu-baseline --no-kernel-leg --no-fp32-leg > gpurun_out/${T}_bench_f32x3.json 2>> gpurun_out/${T}_bench_misc.err
for pr in f32x2 bf16; do timeout 400 python bench.py --model som --points 5000 --nodes 64 --pairs 24 --precision $pr --no-cpu-baseline --no-kernel-leg > gpurun_out/${T}_bench_cfg1_$pr.json 2>> gpurun_out/${T}_bench_misc.err; done
for f in gpurun_out/${T}_bench*.json; do python - "$f" <<'PY'
import json, sys
try:
    ln = [l for l in open(sys.argv[1]) if l.strip().startswith('{')][-1]
    d = json.loads(ln); print(sys.argv[1].split('/')[-1], '%.3f ms/step' % d['ms_per_step'], '%.1f' % d['value'], d.get('unit'), (d.get('roofline') or {}).get('frac'))
except Exception as e:
    print(sys.argv[1], 'unreadable', e)
PY
done
