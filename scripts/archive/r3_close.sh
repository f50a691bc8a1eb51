#!/bin/bash
# closing run of the shipped binary and orders: 1e10 / 1e9-sample soak, the sampled-state contract against the C oracle
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3_close; mkdir -p $O; cd $R
timeout 1500 python scripts/soak2.py 1e10 2>&1 | tee $O/soak.log
timeout 1200 python tests/tools/sampled_vs_oracle.py 1e6 rts24 2>&1 | tee $O/sampled24.log
timeout 1200 python tests/tools/sampled_vs_oracle.py 2e5 rts96 2>&1 | tee $O/sampled96.log
