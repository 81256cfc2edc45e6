#!/bin/bash
# Same-box A/B of the training-step bench under two environments (box-to-box variance is larger than most effects):
#   bash tools/ab_env.sh <tag> "<env A>" "<env B>" [bench args]     e.g.  ... r02r "USIP_TUNE=x3_gemm_tile=3" ""
TAG=$1; A=$2; B=$3; shift 3
mkdir -p gpurun_out
for rep in 1 2; do
  for side in A B; do
    if [ $side = A ]; then E="$A"; else E="$B"; fi
    env $E timeout 300 python bench.py --no-kernel-leg --no-cpu-baseline "$@" 2> gpurun_out/${TAG}_ab_${side}${rep}.err | tail -n 1 > gpurun_out/${TAG}_ab_${side}${rep}.json
    python - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_ab_${side}${rep}.json"))
print("${side}${rep} [%s] ms_per_step %.4f value %.1f" % ("$E", d["ms_per_step"], d["value"]))
PY
  done
done
