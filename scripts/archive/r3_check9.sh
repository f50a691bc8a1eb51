#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3c9; mkdir -p $O; cd $R
python scripts/variant_check.py pm0 base pm0 base 2>&1 | tee $O/v24.log
python scripts/variant_check96.py pm0 base 2>&1 | tee $O/v96.log
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee $O/pytest.log
