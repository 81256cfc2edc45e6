#!/bin/bash
# (the TCC_* counter sets hung rocprofv3 for 15 minutes on this pool: left out; every pass has its own timeout)
# SQ counters of the kernels whose name contains <substr>, averaged per launch:
#   bash tools/pmc_any.sh <tag> <substr> <python script + args>
TAG=$1; SUB=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC" \
           "TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TA_TCP_STATE_READ_sum"; do
  N=$(echo $SET | cut -d' ' -f1)
  timeout 150 rocprofv3 --pmc $SET --output-format csv -d $OUT/$N -o p -- python $ROOT/"$@" > $OUT/$N.log 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "$SUB" in k:
            acc[k[:110]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, d in acc.items():
    print(name)
    for c, v in sorted(d.items()):
        print("   %-32s %14.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
