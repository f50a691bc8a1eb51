#!/bin/bash
# pair reciprocals at the injections' ratio tests on top of the one-reciprocal evaluation: 0xDD (slacks), 0xED (multipliers), 0xFD (both)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3c23; mkdir -p $O; cd $R
python scripts/variant_check.py base mdd med mfd base mdd med mfd 2>&1 | tee $O/v24.log
