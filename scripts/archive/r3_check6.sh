#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3c6; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_host.py tests/test_c_abi.py -m gpu -x -q 2>&1 | tail -30 | tee $O/pytest.log
