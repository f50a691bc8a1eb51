#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3c2; mkdir -p $O; cd $R
python scripts/variant_check.py r2 base v_dmukeep v_zplus v_zplus_dmukeep base 2>&1 | tee $O/v24.log
python scripts/variant_check96.py r2 base v_dmukeep v_zplus v_zplus_dmukeep 2>&1 | tee $O/v96.log
