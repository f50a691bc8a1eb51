#!/bin/bash
# back-substitution passes filled to at most a half in half form (six LDS instructions instead of seven): same binary, RELMC_NO_BWD_HALF=1 = before
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3c25; mkdir -p $O; cd $R
for e in "RELMC_NO_BWD_HALF=1" "RELMC_X=0" "RELMC_NO_BWD_HALF=1" "RELMC_X=0"; do
  echo "== $e" | tee -a $O/v24.log $O/v96.log
  env $e python scripts/variant_check.py base 2>&1 | tee -a $O/v24.log
  env $e python scripts/variant_check96.py base 2>&1 | tee -a $O/v96.log
done
