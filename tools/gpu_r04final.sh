#!/bin/bash
# round-end evidence: the whole GPU suite, then tools/gpu_r04prof.sh (profile, PMC, bench lines)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
T=$1
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > gpurun_out/${T}_pytest.txt
cat gpurun_out/${T}_pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/gpu_r04prof.sh $T
