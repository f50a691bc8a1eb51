#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3c5; mkdir -p $O; cd $R
python tests/tools/nodal96_agg.py 2>&1 | tee $O/nodal96.log
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/pytest.log
python scripts/variant_check.py base 2>&1 | tee $O/v24.log
