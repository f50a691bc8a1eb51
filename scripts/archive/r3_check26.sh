#!/bin/bash
# half-form back substitution (base) against the previous commit's binary (head), same box
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3c26; mkdir -p $O; cd $R
python scripts/variant_check.py head base head base 2>&1 | tee $O/v24.log
python scripts/variant_check96.py head base head base 2>&1 | tee $O/v96.log
