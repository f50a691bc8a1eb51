#!/bin/bash
# priority balancing, further parameters on top of bit 12: high level 2 instead of 3 (ph2), re-evaluated before the Newton step too (pt2; pt2b13: on bit 11)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3c39; mkdir -p $O; cd $R
python scripts/variant_check.py base ph2 pt2 pt2b13 base ph2 pt2 2>&1 | tee $O/v24.log
python scripts/variant_check96.py base ph2 pt2 pt2b13 base 2>&1 | tee $O/v96.log
