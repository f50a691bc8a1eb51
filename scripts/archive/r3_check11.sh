#!/bin/bash
# slot fences of the vector phases: which sites still pay? (RELMC_FENCE_MASK variants f00 none, f03 evaluation only, f0c ratio tests only, f30 update only, f3c all but the evaluation)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3c11; mkdir -p $O; cd $R
python scripts/variant_check.py base f00 f03 f0c f30 f3c base f00 2>&1 | tee $O/v24.log
python scripts/variant_check96.py base f00 f03 f0c f30 f3c base 2>&1 | tee $O/v96.log
