#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/db_trace; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -f csv -d $O/t -o db -- python3 $R/scripts/db_rate.py 1000000 > $O/log.txt 2>&1
python3 - <<PY
import csv, glob
f = glob.glob('$O/t/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    print(r["Name"][:70], r["Calls"], "avg us", round(float(r["AverageNs"])/1e3,1), "total ms", round(float(r["TotalDurationNs"])/1e6,2))
PY
