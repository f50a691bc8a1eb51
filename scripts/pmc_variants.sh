#!/bin/bash
# PMC passes of one 1e6-scenario launch for several library builds:  bash scripts/pmc_variants.sh <lib in csrc/ablate> ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; cd /tmp
for V in "$@"; do
  OUT=$R/gpurun_out/pmcv_$V; mkdir -p $OUT
  if [ "$V" != "base" ]; then export RELMC_LIB_PATH=$R/powersystemsreliabilityassessment_amd/csrc/ablate/$V.so; else unset RELMC_LIB_PATH; fi
  CMD="python3 $R/scripts/one_launch.py 1000000"
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -f csv -d $OUT/p1 -o pmc -- $CMD > $OUT/p1.log 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU -f csv -d $OUT/p3 -o pmc -- $CMD > $OUT/p3.log 2>&1
  python3 - <<PY
import csv, glob
for p in ('p1','p3'):
    f = glob.glob('$OUT/'+p+'/**/*counter_collection.csv', recursive=True)
    if not f: print('$V', p, 'no csv'); continue
    rows = [r for r in csv.DictReader(open(f[0])) if 'eval_kernel' in r['Kernel_Name']]
    last = max(int(r['Dispatch_Id']) for r in rows)
    print('$V', p, {r['Counter_Name']: float(r['Counter_Value']) for r in rows if int(r['Dispatch_Id'])==last})
PY
  grep kernel_ms $OUT/p1.log
done
