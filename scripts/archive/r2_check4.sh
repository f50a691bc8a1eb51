#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r2c4; mkdir -p $O; cd $R
python scripts/variant_check96.py base t96_1wave 2>&1 | tee $O/v96.log
python -m pytest tests/test_rts96.py tests/test_gpu_parity.py -m gpu -x -q -k "numfail or nonconverged or golden_within" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -8 $O/pytest.log
