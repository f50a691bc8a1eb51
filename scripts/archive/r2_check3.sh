#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r2c3; mkdir -p $O; cd $R
python scripts/pcie_rate.py > $O/pcie.log 2>&1; cat $O/pcie.log
python -m pytest tests/ -m gpu -x -q > $O/pytest_all.log 2>&1; echo "pytest rc $?" >> $O/pytest_all.log; tail -8 $O/pytest_all.log
