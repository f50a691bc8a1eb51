#!/bin/bash
# step-phase solution batch: one batch (base, 208 B/lane scratch), two batches (sp, 192), none (c0, 176): time and HBM traffic
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3c22; mkdir -p $O; cd $R
python scripts/variant_check.py base sp c0 base sp c0 2>&1 | tee $O/v24.log
bash scripts/traffic.sh base sp c0 2>&1 | tee $O/traffic.log
