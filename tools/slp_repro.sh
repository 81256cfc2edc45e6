#!/bin/bash
# Reproducer of the packed-fp32 failure described in DESIGN.md 5 / usip_amd/build.py: the fused layer backward compiled
# WITH hipcc's SLP vectoriser (the library's flags minus -fno-slp-vectorize), linked against the library's other objects,
# and launched repeatedly on the same inputs.  Step 1 (CPU, here): bash tools/slp_repro.sh build  -> gpurun_out/libusip_slp.so
# Step 2 (GPU box, through gpurun):  bash tools/slp_repro.sh run [N]
# Expected with ROCm 7.2 on gfx950: the 64 -> 128 rows report most launches differing (1-4 wrong tiles of 1024, errors of
# ~1e-3 on 16 positions of one tile), every other row and the shipped library 0.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
case "${1:-build}" in
  build)
    python -m usip_amd.build > /dev/null
    mkdir -p gpurun_out
    FLAGS=$(python -c "from usip_amd.build import FLAGS; print(' '.join(f for f in FLAGS if f != '-fno-slp-vectorize'))")
    /opt/rocm/bin/hipcc $FLAGS -x hip -c usip_amd/csrc/layer_bwd_x2.hip -o gpurun_out/layer_bwd_x2_slp.o
    OBJS=$(ls usip_amd/build/*.o | grep -v layer_bwd_x2)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_out/libusip_slp.so $OBJS gpurun_out/layer_bwd_x2_slp.o -lpthread
    rm -f gpurun_out/layer_bwd_x2_slp.o
    echo "built gpurun_out/libusip_slp.so (note: gpurun_out/ does not travel to the GPU box; copy it next to the library first)"
    ;;
  run)
    N=${2:-60}
    LIB=${USIP_SLP_LIB:-$ROOT/usip_amd/libusip_slp.so}
    echo "== with the SLP vectoriser ($LIB)"
    USIP_LIB=$LIB python tools/layer_bwd_race.py $N 2>&1 | grep -v "differs:"
    echo "== the shipped library"
    python tools/layer_bwd_race.py $N 2>&1 | grep -v "differs:"
    ;;
esac
