#!/bin/bash
# round 3, first GPU call: reciprocal variants (same box), LDS atomic microbenchmark
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3c1; mkdir -p $O; cd $R
python scripts/variant_check.py r2 v_rcp3 v_pair v_keep_nopair base r2 base 2>&1 | tee $O/v24.log
python scripts/variant_check96.py r2 v_rcp3 v_pair v_keep_nopair base 2>&1 | tee $O/v96.log
./scripts/micro/lds_atomic 2>&1 | tee $O/lds_atomic.txt
