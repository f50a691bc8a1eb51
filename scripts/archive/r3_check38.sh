#!/bin/bash
# priority balancing: a coin per wavefront and iteration (pbr) against clock bits 12 / 18 and the shipped 15
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3c38; mkdir -p $O; cd $R
python scripts/variant_check.py base pbr pb12 pb18 base pbr pb12 2>&1 | tee $O/v24.log
python scripts/variant_check96.py base pbr pb12 pb18 base pbr pb12 2>&1 | tee $O/v96.log
