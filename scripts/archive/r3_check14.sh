#!/bin/bash
# prefetch-site combinations per tile (q78: 16-lane 0x78 / wide 0x40; q48: 0x48 / 0x42; q58: 0x58 / 0x46; q40: 0x40 / 0x48)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r3c14; mkdir -p $O; cd $R
python scripts/variant_check.py base q78 q48 q58 q40 base q78 q48 2>&1 | tee $O/v24.log
python scripts/variant_check96.py base q78 q48 q58 q40 base q78 q48 2>&1 | tee $O/v96.log
