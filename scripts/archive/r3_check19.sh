#!/bin/bash
# pass descriptors as byte offsets (base) against offsets in doubles (head = the previous commit)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3c19; mkdir -p $O; cd $R
python scripts/variant_check.py head base head base 2>&1 | tee $O/v24.log
python scripts/variant_check96.py head base head base 2>&1 | tee $O/v96.log
