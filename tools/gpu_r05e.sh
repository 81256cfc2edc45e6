#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_f32x2_mode_gpu.py -m gpu -q -x -k "full_line or wgrad" > gpurun_out/r05e_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r05e_tests.log; tail -3 gpurun_out/r05e_tests.log
for i in 1 2; do
for w in 0 1; do python tools/wgrad_pooled_one.py 512 512 8192 16 r5_forms $w; python tools/wgrad_one.py 512 256 8192 16 r5_forms $w; python tools/wgrad_one.py 256 256 8192 16 r5_forms $w; done; done 2>&1 | tee gpurun_out/r05e_wgrad_times.txt
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_r05e -o p -- python $GRAFT_REPO_ROOT/tools/wgrad_pooled_one.py > /dev/null 2>&1
python - <<P
import csv,glob,collections,os
acc=collections.defaultdict(list)
for f in glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/pmc_r05e/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "wgrad_x2l" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({k:sum(v)/len(v) for k,v in acc.items()})
P
cd $GRAFT_REPO_ROOT
for f in 0 1 0 1; do
  timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-fp32-leg --no-kernel-leg --tune r5_forms=$f > gpurun_out/r05e_ab_$f.json 2>> gpurun_out/r05e_ab.err
  python - <<P
import json
d=json.loads(open("gpurun_out/r05e_ab_$f.json").read().strip().splitlines()[-1])
print("r5_forms=$f  ms=%.4f median=%.4f"%(d["ms_per_step"], d["step_ms_rank0"]["median"]))
P
done 2>&1 | tee gpurun_out/r05e_ab.txt
rm -rf gpurun_out/pmc_r05e
