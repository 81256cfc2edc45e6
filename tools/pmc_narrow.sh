#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_narrow
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"; do
  N=$(echo $SET | cut -d' ' -f1)
  rocprofv3 --pmc $SET --output-format csv -d $OUT/$N -o p -- python $ROOT/tools/narrow_bwd_bench.py > $OUT/$N.log 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "narrow_bwd" in k:
            acc["narrow 128" if "Li128" in k or "<128" in k else "narrow 64"][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, d in acc.items():
    print(name)
    for c, v in sorted(d.items()):
        print("   %-32s %14.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
