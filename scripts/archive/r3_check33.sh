#!/bin/bash
# selection among the tuned RTS-24 candidate orders: time, fixture iteration counts, nodal sums against the oracle
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3c33; mkdir -p $O; cd $R
python tests/tools/order_select.py rts24 rule scripts/orders/cand24/*.txt scripts/orders/24_c3.txt scripts/orders/24_t1.txt rule 2>&1 | tee $O/select24.log
