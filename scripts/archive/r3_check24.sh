#!/bin/bash
# the shipped arithmetic against the C oracle on 1e6 sampled RTS-24 states and 2e5 sampled RTS-96 states (both policies)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3c24; mkdir -p $O; cd $R
timeout 1200 python tests/tools/sampled_vs_oracle.py 1e6 rts24 2>&1 | tee $O/sampled24.log
timeout 1200 python tests/tools/sampled_vs_oracle.py 2e5 rts96 2>&1 | tee $O/sampled96.log
