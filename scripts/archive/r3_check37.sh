#!/bin/bash
# priority balancing: which bit of the shader clock the first-dispatched wavefronts alternate on (shipped until now: 15), or the iteration parity (pbm1)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3c37; mkdir -p $O; cd $R
python scripts/variant_check.py base pb9 pb10 pb11 pb12 pb18 pb19 pb20 pb22 pbm1 base pb12 pb10 pb11 2>&1 | tee $O/v24b.log
python scripts/variant_check96.py base pb9 pb10 pb11 pb12 pb18 pb19 pb20 pb22 pbm1 base pb12 2>&1 | tee $O/v96b.log
