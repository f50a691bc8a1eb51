#!/bin/bash
# wide tile with the one-reciprocal injection evaluation + pair mask as shipped arithmetic: time, fixture, rescan of the primary order's failing states
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3c42; mkdir -p $O; cd $R
python scripts/variant_check96.py base base 2>&1 | tee $O/v96.log
RELMC_NO_RETRY=1 python tests/tools/numfail96.py 1e8 1 > $O/numfail96_scan.json 2> $O/scan.err; tail -1 $O/scan.err
python scripts/order_soak.py rts96 2e8 2>&1 | tee $O/soak.log
