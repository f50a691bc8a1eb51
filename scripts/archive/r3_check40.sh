#!/bin/bash
# wide tile: injection evaluation from one reciprocal (nw), plus the 16-lane tile's pair mask (nwp)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3c40; mkdir -p $O; cd $R
python scripts/variant_check96.py base nw nwp base nw nwp 2>&1 | tee $O/v96.log
