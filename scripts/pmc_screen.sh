#!/bin/bash
# PMC passes on the pre-screen's kernels (sample / flag kernel of the last screened launch):  bash scripts/pmc_screen.sh <tag> [rts24|rts96|seq]
TAG=${1:-x}; W=${2:-rts96}; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/pmcs_${TAG}_$W; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
CMD="python3 $R/scripts/screen_launch.py $W"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 -f csv -d $OUT/p1 -o pmc -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INST_CYCLES_VMEM_RD -f csv -d $OUT/p2 -o pmc -- $CMD > $OUT/p2.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 GRBM_GUI_ACTIVE -f csv -d $OUT/p3 -o pmc -- $CMD > $OUT/p3.log 2>&1
python3 - <<PY
import csv, glob
for p in ('p1','p2','p3'):
    f = glob.glob('$OUT/'+p+'/*counter_collection.csv')
    if not f: print(p, 'no csv'); continue
    rows = [r for r in csv.DictReader(open(f[0])) if 'screen_sample_kernel' in r['Kernel_Name'] or 'seq_flag_kernel' in r['Kernel_Name']]
    if not rows: print(p, 'no rows'); continue
    last = max(int(r['Dispatch_Id']) for r in rows)
    print(p, {r['Counter_Name']: float(r['Counter_Value']) for r in rows if int(r['Dispatch_Id'])==last})
PY
