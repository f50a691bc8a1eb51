#!/bin/bash
# Instruction-class counters of ONE launch of the fused kernel (what scripts/isa_budget.py reconciles its static budget with):
#   bash scripts/pmc_classes.sh <tag> [rts24|rts96] [samples]      -> gpurun_out/pmc_classes_<tag>_<case>.json
# Two rocprofv3 --pmc passes (counters only: no trace domains), the program directly after `--`.
#   bash scripts/pmc_classes.sh <tag> <case> <samples> <max_it>   -> ..._<case>_it<max_it>.json: every row stops after max_it Newton steps (one_launch.py)
TAG=${1:-x}; CASE=${2:-rts24}; N=${3:-1000000}; MAXIT=${4:-}; SFX=${MAXIT:+_it$MAXIT}; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/pmc_classes_${TAG}_$CASE$SFX; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
CMD="python3 $R/scripts/one_launch.py $N $CASE $MAXIT"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT -f csv -d $OUT/a -o pmc -- $CMD > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -f csv -d $OUT/b -o pmc -- $CMD > $OUT/b.log 2>&1
python3 - <<PY
import csv, glob, json
res = {}
for p in ('a', 'b'):
    f = glob.glob('$OUT/' + p + '/**/*counter_collection.csv', recursive=True)
    if not f: print(p, 'no csv'); continue
    rows = [r for r in csv.DictReader(open(f[0])) if 'eval_kernel<0' in r['Kernel_Name']]
    last = max(int(r['Dispatch_Id']) for r in rows)          # the measured launch (the first one is the case load's order probe, the second a warm-up)
    res.update({r['Counter_Name']: float(r['Counter_Value']) for r in rows if int(r['Dispatch_Id']) == last})
    res['kernel'] = [r['Kernel_Name'] for r in rows if int(r['Dispatch_Id']) == last][0].split('(')[0]
for ln in open('$OUT/a.log'):
    if ln.startswith('kernel_ms'):
        t = ln.split(); res.update(kernel_ms=float(t[1]), mean_iters=float(t[5]), mean_max_iters=float(t[7]), samples=$N, case='$CASE')
json.dump(res, open('$R/gpurun_out/pmc_classes_${TAG}_$CASE$SFX.json', 'w'), indent=1)
print(json.dumps(res))
PY
