#!/bin/bash
# the RTS-96 candidate orders against the pins: time, fixture iteration counts, nodal sums against the oracle (3e5 sampled states)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3c34; mkdir -p $O; cd $R
python tests/tools/order_select.py rts96 rule scripts/orders/96_c1.txt scripts/orders/96_t3.txt scripts/orders/96_t5.txt 2>&1 | tee $O/select96.log
