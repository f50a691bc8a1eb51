#!/bin/bash
# tuned primary elimination orders (scripts/order_tune.py against the real scheduler): orders in $R/scripts/orders/<case>_<name>.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3c28; mkdir -p $O; cd $R
unset RELMC_ORDER
python scripts/variant_check96.py base 2>&1 | tee -a $O/v96.log
for f in $R/scripts/orders/96_*.txt; do
  echo "== $(basename $f)" | tee -a $O/v96.log
  RELMC_ORDER=$(cat $f) python scripts/variant_check96.py base 2>&1 | tee -a $O/v96.log
done
python scripts/variant_check96.py base 2>&1 | tee -a $O/v96.log
python scripts/variant_check.py base 2>&1 | tee -a $O/v24.log
for f in $R/scripts/orders/24_*.txt; do
  echo "== $(basename $f)" | tee -a $O/v24.log
  RELMC_ORDER=$(cat $f) python scripts/variant_check.py base base 2>&1 | tee -a $O/v24.log
done
python scripts/variant_check.py base 2>&1 | tee -a $O/v24.log
