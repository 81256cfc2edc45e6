#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats + HBM traffic counters of the bench step.
# Counters are collected in their OWN passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950,
# MI355X_MICROARCH.md "rocprofv3 PMC slots") and never together with sys/runtime tracing.
#   usage: tools/profile_roofline.sh <tag> [extra bench.py flags]   -> gpurun_out/prof_<tag>/{trace,fetch,write}/...
set -u
TAG=${1:-r01}
shift || true
EXTRA="$*"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-graph --no-fp32-leg $EXTRA"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $CMD > "$OUT/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o fetch -- $CMD > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o write -- $CMD > "$OUT/write.log" 2>&1
# the stand-alone leg of the two HBM-bound integer kernels (dist-in ball_query, index_max): same three passes
KCMD="python $ROOT/bench.py --only-kernels"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o ktrace -- $KCMD > "$OUT/ktrace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o kfetch -- $KCMD > "$OUT/kfetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o kwrite -- $KCMD > "$OUT/kwrite.log" 2>&1
find "$OUT" -name "*.csv" | head -20
python "$ROOT/tools/pmc_summary.py" "$OUT" "$OUT/summary_$TAG.txt" "$OUT/traffic_$TAG.json"
