#!/bin/bash
# prefetch-site combinations, second round (16-lane / wide): r1 0x48 / 0x46, r2 0xc8 / 0xc6, r3 0x08 / 0x06, r4 0x4c / 0x4e, r5 0x68 / 0x56, r6 0x49 / 0x66
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3c15; mkdir -p $O; cd $R
python scripts/variant_check.py base r1 r2 r3 r4 r5 r6 base r1 r2 2>&1 | tee $O/v24.log
python scripts/variant_check96.py base r1 r2 r3 r4 r5 r6 base r1 r2 2>&1 | tee $O/v96.log
