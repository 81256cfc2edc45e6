#!/bin/bash
# bit-reproducibility soak of the final build: three training steps from one seed, repeated; every parameter compared bit for bit
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
{ timeout 600 python tools/repro_stress.py ball 60
  timeout 600 python tools/repro_stress.py som 40
  timeout 900 python tools/repro_stress.py ball 25 8 16384 512
  timeout 600 python tools/repro_stress.py som 15 8 16384 512; } > gpurun_out/r06an_repro.txt 2>&1
cat gpurun_out/r06an_repro.txt | grep -v amdgpu.ids
