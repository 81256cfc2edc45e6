#!/bin/bash
# reproducibility soak of the final build (tools/repro_stress.py: three Adam steps from one seed, every parameter bit compared)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
{
timeout 900 python tools/repro_stress.py ball 150 8 16384 512
timeout 300 python tools/repro_stress.py som 30 8 16384 512
timeout 300 python tools/repro_stress.py knn 40
timeout 300 python tools/repro_stress.py lite 40
USIP_DEFER_WGRAD=0 timeout 300 python tools/repro_stress.py ball 30 8 16384 512
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05i_soak.txt
timeout 900 python -m pytest tests/test_f32x2_mode_gpu.py -m gpu -q -k "bit_stab or repeated or stable or soak" 2>&1 | tail -2 | tee -a gpurun_out/r05i_soak.txt
