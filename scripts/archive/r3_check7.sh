#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3c7; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_rts96.py -m gpu -x -q -k "dense or numfail or retry" 2>&1 | tail -30 | tee $O/pytest.log
