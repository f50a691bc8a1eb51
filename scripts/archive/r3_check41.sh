#!/bin/bash
# the sequential instantiation (MODE 2) under other prefetch masks of the 16-lane tile (shipped 0xc0)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3c41; mkdir -p $O; cd $R
for v in base s00 sc8 sc6 base s00 sc8 sc6; do
  if [ "$v" != "base" ]; then export RELMC_LIB_PATH=$R/powersystemsreliabilityassessment_amd/csrc/ablate/$v.so; else unset RELMC_LIB_PATH; fi
  python bench.py --workload seq --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(b['value']/1e6,2), 'M/s  ms/step', round(b['ms_per_step'],3), 'kernel', round(b['roofline']['kernel_ms_avg'],3))" | tee -a $O/seq.log
done
