#!/bin/bash
# placement search: weight of the written operands' conflicts (RELMC_PLACE_WW) and search length (RELMC_PLACE_MOVES), same binary
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3c21; mkdir -p $O; cd $R
for e in "" "RELMC_PLACE_WW=2" "RELMC_PLACE_WW=3" "RELMC_PLACE_WW=5" "RELMC_PLACE_MOVES=1000" "RELMC_PLACE_MOVES=1000 RELMC_PLACE_WW=3" "RELMC_PLACE_MOVES=0" ""; do
  echo "== $e" | tee -a $O/v24.log $O/v96.log
  env $e python scripts/variant_check.py base 2>&1 | tee -a $O/v24.log
  env $e python scripts/variant_check96.py base 2>&1 | tee -a $O/v96.log
done
