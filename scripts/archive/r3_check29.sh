#!/bin/bash
# how often does the primary order fail under the tuned elimination orders?  (2e8 RTS-96 / 2e9 RTS-24 samples each, both policies)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3c29; mkdir -p $O; cd $R
unset RELMC_ORDER
echo "== rule" | tee -a $O/soak.log; python scripts/order_soak.py rts96 2e8 2>&1 | tee -a $O/soak.log
for k in t3 t5 o1; do echo "== 96_$k" | tee -a $O/soak.log; RELMC_ORDER=$(cat scripts/orders/96_$k.txt) python scripts/order_soak.py rts96 2e8 2>&1 | tee -a $O/soak.log; done
echo "== 24_t1" | tee -a $O/soak.log; RELMC_ORDER=$(cat scripts/orders/24_t1.txt) python scripts/order_soak.py rts24 2e9 2>&1 | tee -a $O/soak.log
