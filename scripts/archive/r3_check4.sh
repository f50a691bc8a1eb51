#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3c4; mkdir -p $O; cd $R
for s in 1 2 4; do echo "RELMC_SPLIT_MAX=$s"; RELMC_SPLIT_MAX=$s python scripts/variant_check96.py base; done 2>&1 | tee $O/v96.log
for s in 1 2; do echo "RELMC_SPLIT_MAX=$s"; RELMC_SPLIT_MAX=$s python scripts/variant_check.py base; done 2>&1 | tee $O/v24.log
