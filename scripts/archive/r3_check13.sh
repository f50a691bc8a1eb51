#!/bin/bash
# LDS round trips of the vector phases off the critical path, per site (RELMC_PF_MASK): pf01 .. pf40 single sites, pf7f all
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3c13; mkdir -p $O; cd $R
python scripts/variant_check.py base pf01 pf02 pf04 pf08 pf10 pf20 pf40 pf7f base pf7f 2>&1 | tee $O/v24.log
python scripts/variant_check96.py base pf01 pf02 pf04 pf08 pf10 pf20 pf40 pf7f base 2>&1 | tee $O/v96.log
