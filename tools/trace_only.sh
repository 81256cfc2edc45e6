#!/bin/bash
# kernel trace of the bench step only (no counters): tools/trace_only.sh <tag> [bench flags] -> gpurun_out/trace_<tag>.txt
TAG=${1:-t}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-graph "$@" > "$OUT/trace.log" 2>&1
python "$ROOT/tools/pmc_summary.py" "$OUT" "$ROOT/gpurun_out/trace_$TAG.txt" "$OUT/traffic.json" 2>&1 | tail -2
