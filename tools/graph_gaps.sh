#!/bin/bash
# kernel time against wall time of the graph-replayed step: rocprofv3 --kernel-trace of bench.py (graphs on), then the
# dispatches between the first and the last kernel of every replay: busy time / span
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/gaps
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o g -- python $ROOT/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-kernel-leg > $OUT/run.log 2>&1
tail -c 600 $OUT/run.log
python - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
# the timed region: the last 8 x N dispatches; find step boundaries by the adam kernel
idx = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
print("adam launches:", len(idx))
for a, b in list(zip(idx[:-1], idx[1:]))[-6:]:
    seg = rows[a + 1:b + 1]
    span = seg[-1][1] - seg[0][0]
    busy = sum(e - s for s, e, _ in seg)
    gaps = sorted(((seg[i + 1][0] - seg[i][1]), seg[i][2][:40], seg[i + 1][2][:40]) for i in range(len(seg) - 1))
    print("launches %d  span %.1f us  kernel time %.1f us  idle %.1f us  largest gaps: %s" % (
        len(seg), span / 1e3, busy / 1e3, (span - busy) / 1e3, [(round(g[0] / 1e3, 1), g[1], g[2]) for g in gaps[-3:]]))
PY
