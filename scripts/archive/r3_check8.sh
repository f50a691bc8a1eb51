#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3c8; mkdir -p $O; cd $R
timeout 900 python tests/tools/fuzz_cases.py 40 2>&1 | tee $O/fuzz.log | tail -45
timeout 900 python scripts/retry_soak.py 1000000000 2>&1 | tee $O/soak.log
