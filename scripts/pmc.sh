#!/bin/bash
# PMC passes on one 1e6-scenario launch:  bash scripts/pmc.sh <tag>
TAG=${1:-x}; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/pmc_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
CMD="python3 $R/scripts/one_launch.py 1000000"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VMEM SQ_INSTS_LDS -f csv -d $OUT/p1 -o pmc -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_INST_CYCLES_VMEM_RD -f csv -d $OUT/p2 -o pmc -- $CMD > $OUT/p2.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM -f csv -d $OUT/p3 -o pmc -- $CMD > $OUT/p3.log 2>&1
python3 - <<PY
import csv, glob, collections
for p in ('p1','p2','p3'):
    f = glob.glob('$OUT/'+p+'/*counter_collection.csv')
    if not f: print(p, 'no csv'); continue
    rows = [r for r in csv.DictReader(open(f[0])) if 'eval_kernel<0' in r['Kernel_Name']]
    last = max(int(r['Dispatch_Id']) for r in rows)
    print(p, {r['Counter_Name']: float(r['Counter_Value']) for r in rows if int(r['Dispatch_Id'])==last})
PY
grep kernel_ms $OUT/p1.log
