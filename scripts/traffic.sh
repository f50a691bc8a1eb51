#!/bin/bash
# HBM traffic of one 1e6-scenario launch (FETCH_SIZE / WRITE_SIZE in separate passes):  bash scripts/traffic.sh [lib in csrc/ablate]
R=${GRAFT_REPO_ROOT:-$(pwd)}; export TMPDIR=/tmp; cd /tmp
for V in "$@"; do
  OUT=$R/gpurun_out/traffic_$V; mkdir -p $OUT
  if [ "$V" != "base" ]; then export RELMC_LIB_PATH=$R/powersystemsreliabilityassessment_amd/csrc/ablate/$V.so; else unset RELMC_LIB_PATH; fi
  CMD="python3 $R/scripts/one_launch.py 1000000"
  rocprofv3 --pmc FETCH_SIZE -f csv -d $OUT/f -o pmc -- $CMD > $OUT/f.log 2>&1
  rocprofv3 --pmc WRITE_SIZE -f csv -d $OUT/w -o pmc -- $CMD > $OUT/w.log 2>&1
  python3 - <<PY
import csv, glob
tot = {}
for p in ('f','w'):
    f = glob.glob('$OUT/'+p+'/**/*counter_collection.csv', recursive=True)
    rows = [r for r in csv.DictReader(open(f[0])) if 'eval_kernel' in r['Kernel_Name']]
    last = max(int(r['Dispatch_Id']) for r in rows)
    for r in rows:
        if int(r['Dispatch_Id'])==last: tot[r['Counter_Name']] = float(r['Counter_Value'])
print('$V', tot, 'bytes per scenario', (2*tot['FETCH_SIZE']+tot['WRITE_SIZE'])*1024/1e6)
PY
  grep kernel_ms $OUT/f.log
done
