#!/bin/bash
# robustness of the shipped arithmetic after the injection-evaluation change: fuzz cases against the C oracle, 2e9 RTS-24 + 2e8 RTS-96 samples
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3c18; mkdir -p $O; cd $R
timeout 900 python tests/tools/fuzz_cases.py 40 2>&1 | tee $O/fuzz.log | tail -4
timeout 900 python scripts/soak2.py 2e9 2>&1 | tee $O/soak.log
