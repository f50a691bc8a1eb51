#!/bin/bash
# per-bus nodal sums against the C oracle on 3e5 sampled RTS-24 states under the candidate orders
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3c32; mkdir -p $O; cd $R
for k in rule scripts/orders/24_c3.txt scripts/orders/24_c2.txt scripts/orders/24_t1.txt; do
  echo "== $k" | tee -a $O/nodal.log
  timeout 600 python tests/tools/sampled_vs_oracle.py 3e5 rts24 $k 2>&1 | tee -a $O/nodal.log
done
