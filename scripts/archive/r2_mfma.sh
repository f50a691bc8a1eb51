#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r2_mfma; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
$R/scripts/micro/mfma_kkt 1000000 > $O/mfma_kkt.txt 2>&1; cat $O/mfma_kkt.txt
rocprofv3 --kernel-trace --stats -f csv -d $O/trace -o mfma -- $R/scripts/micro/mfma_kkt 1000000 > $O/trace.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE -f csv -d $O/pmc -o mfma -- $R/scripts/micro/mfma_kkt 1000000 > $O/pmc.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob('$O/pmc/**/*counter_collection.csv', recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if 'mfma_kkt' in r['Kernel_Name']]
last = max(int(r['Dispatch_Id']) for r in rows)
print('PMC last dispatch', {r['Counter_Name']: float(r['Counter_Value']) for r in rows if int(r['Dispatch_Id'])==last})
f = glob.glob('$O/trace/**/*kernel_stats.csv', recursive=True)
print(open(f[0]).read())
PY
cd $R; python scripts/variant_check.py r2base stash r2base stash
