#!/bin/bash
# full-form update pass: all eight operand loads in one batch (base) against the second rows requested after the first row's store (two)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3c17; mkdir -p $O; cd $R
python scripts/variant_check.py two base two base 2>&1 | tee $O/v24.log
python scripts/variant_check96.py two base two base 2>&1 | tee $O/v96.log
